#!/usr/bin/env python3
"""Golden vectors for the attention oracle from the ONE pure-PyTorch attention statement that lives under /root/reference:
`ref_mha_bmhk` / `ref_mha_bmk` of the vendored CUTLASS example
(/root/reference/pod_attn/csrc/cutlass/examples/41_fused_multi_head_attention/fmha_backward_test.py:78-105) —
fp32 softmax(q.k^T / sqrt(d) + mask).v in [batch, rows, heads, d] layout, with its log-sum-exp.

What this pins and what it does not.  It is NOT the operator's own test (the reference holds none for flash_attn_with_kvcache:
SURVEY §8c; oracle/attn.py stays "parity unpinned" in the task's sense).  It is an independent statement of the same arithmetic that
ships inside the reference tree, executed HERE from the reference's file (the two function definitions are compiled out of that file's
AST; the script part of the file — argparse, a CUDA executable — is not run, nothing is copied into this repository), on seeded inputs:
  * whole-prompt causal attention (rows == keys: the example's own lower-triangular mask, fmha_backward_test.py:70-75),
  * no mask,
  * a chunk on a cached prefix and a single decode row, with the BOTTOM-RIGHT aligned mask of the operator
    (flash_attn_interface.py:1168-1254) handed to the function as its `mask` argument — there the function pins softmax.V, the mask
    is ours.
Round 4 adds the OPERATOR BY COMPOSITION (OP_CASES below): append by the reference's own cache_flat statement, select by cache_batch_idx,
cut at cache_seqlens, attend by ref_mha_bmhk — the index semantics of flash_attn_with_kvcache pinned against reference text as far as
the tree allows (no operator fixture exists: the row stays "partial", stated in oracle/attn.py).
The outputs (fp32) and LSEs are stored with the inputs' seeds in tests/golden/attn_intree_ref_mha.npz; tests/test_attn_oracle.py
replays them against oracle/attn.py on every run (the file travels; /root/reference does not).
usage: python oracle/gen_golden_attn_intree.py"""
import ast
import os

import numpy as np
import torch

REF = "/root/reference/pod_attn/csrc/cutlass/examples/41_fused_multi_head_attention/fmha_backward_test.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "attn_intree_ref_mha.npz")
WANTED = ("ref_mha_bmk", "bmhk2bmk", "ref_mha_bmhk")

# (name, batch, query rows, visible keys, query heads, kv heads, d, causal, dtype)
CASES = [
    ("causal_whole_prompt", 1, 96, 96, 2, 2, 128, True, "float16"),
    ("causal_whole_prompt_gqa", 1, 130, 130, 4, 2, 128, True, "bfloat16"),
    ("no_mask", 2, 33, 77, 2, 2, 64, False, "float16"),
    ("chunk_on_prefix_bottom_right", 1, 64, 200, 4, 1, 128, True, "float16"),
    ("decode_row_bottom_right", 3, 1, 257, 8, 4, 128, True, "float16"),
]


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED), "the reference file no longer defines " + str(WANTED)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)      # the reference's own code objects
    return ns["ref_mha_bmhk"]


def inputs(seed, B, Sq, Sk, Hq, Hkv, D, dtype):
    g = torch.Generator().manual_seed(seed)
    dt = getattr(torch, dtype)
    q = torch.randn(B, Sq, Hq, D, generator=g).to(dt)
    k = torch.randn(B, Sk, Hkv, D, generator=g).to(dt)
    v = torch.randn(B, Sk, Hkv, D, generator=g).to(dt)
    return q, k, v


# ---- the OPERATOR by composition (round 4) --------------------------------------------------------------------------------------
# flash_attn_with_kvcache = append, select the slot, cut at the length, attend.  Each step but the last has a statement inside the
# reference tree that can be executed here:
#   append   sarathi-lean/csrc/cache_kernels.cu:483-520 (cache_flat_kernel): k_cache[t * k_cache_stride + i] = key[t * key_stride + i]
#            for t < num_tokens, i < num_heads * head_size — restated below as that flat index arithmetic (numpy, no slicing sugar)
#            on the slot's rows from `cache_seqlens[b]` on, which is where the wrapper points it
#            (vattention_flashattention_wrapper.py:151-156) and where the operator's own append writes (flash_attn_interface.py:1168-1176);
#   select   cache_batch_idx[b] names the cache row-block of batch entry b (flash_attn_interface.py:1216-1219);
#   cut      keys [0, cache_seqlens[b] + seqlen_new) are visible (block_info.h:22-23);
#   attend   ref_mha_bmhk of the vendored CUTLASS example on exactly those rows, kv head h // (Hq / Hkv) for query head h, the
#            bottom-right aligned causal mask as its `mask` argument (mask.h:164-196).
# What stays OURS in this composition: the mask's alignment, the GQA mapping, "a row that sees no key is 0" (ref_mha_bmhk yields NaN
# there; stored as the operator's 0 and flagged in `masked_rows`).  Everything else is executed from reference text.
# (name, B, Sq, Sn new tokens (0 = cache already holds them), cached lengths per entry, slots in the cache, cache_batch_idx or None,
#  Hq, Hkv, D, causal, dtype)
OP_CASES = [
    ("decode_append_g4", 3, 1, 1, [0, 62, 63], 5, [4, 0, 2], 8, 2, 128, True, "float16"),
    ("decode_append_g7", 2, 1, 1, [64, 4096], 3, [2, 1], 7, 1, 128, True, "float16"),
    ("decode_append_g8_bf16", 2, 1, 1, [199, 5], 4, [3, 0], 8, 1, 128, True, "bfloat16"),
    ("decode_append_d64", 3, 1, 1, [63, 64, 0], 3, None, 4, 1, 64, True, "float16"),
    ("decode_no_append_lk1", 2, 1, 0, [1, 65], 4, [1, 3], 4, 4, 128, True, "float16"),
    ("decode_no_append_noncausal", 2, 1, 0, [63, 64], 2, None, 8, 2, 64, False, "bfloat16"),
    ("decode_lk4097", 1, 1, 1, [4096], 2, [1], 8, 2, 128, True, "float16"),
    ("chunk_first_g4", 1, 33, 33, [0], 2, [1], 4, 1, 128, True, "float16"),
    ("chunk_on_prefix_g7", 1, 20, 20, [45], 3, [2], 7, 1, 128, True, "float16"),
    ("chunk_on_prefix_d64_bf16", 1, 31, 31, [33], 2, None, 8, 2, 64, True, "bfloat16"),
    ("chunk_prefilled_g8", 2, 17, 0, [64, 65], 4, [0, 3], 8, 1, 128, True, "float16"),
    ("chunk_tile_edges", 3, 5, 5, [58, 59, 60], 3, [2, 0, 1], 4, 2, 128, True, "float16"),
    ("chunk_longer_than_keys", 2, 9, 0, [4, 9], 2, None, 4, 2, 128, True, "float16"),          # rows that see NO key (Sq > Lk): output 0
    ("chunk_noncausal", 2, 6, 6, [10, 0], 3, [1, 2], 4, 4, 64, False, "float16"),
    ("chunk_lk4097", 1, 3, 3, [4094], 1, None, 4, 1, 128, True, "float16"),
    ("one_token_chunk", 2, 1, 0, [1, 64], 2, [1, 0], 8, 2, 128, True, "bfloat16"),
    ("mha_decode", 2, 1, 1, [100, 7], 2, None, 3, 3, 64, True, "float16"),
    ("ragged_batch_g8", 4, 1, 1, [0, 1, 63, 300], 6, [5, 1, 0, 3], 8, 1, 128, True, "float16"),
    ("strided_view", 2, 1, 1, [30, 70], 3, [2, 0], 8, 2, 128, True, "float16"),                # caches handed over as [:, :max_len] views
    ("chunk_g4_d128_bf16", 1, 24, 24, [40], 2, [0], 8, 2, 128, True, "bfloat16"),
]


def op_inputs(seed, B, Sq, Sn, lens, slots, Hq, Hkv, D, dtype):
    g = torch.Generator().manual_seed(seed)
    dt = getattr(torch, dtype)
    smax = max(lens) + max(Sn, 1) + 3
    kc = torch.randn(slots, smax, Hkv, D, generator=g).to(dt)
    vc = torch.randn(slots, smax, Hkv, D, generator=g).to(dt)
    q = torch.randn(B, Sq, Hq, D, generator=g).to(dt)
    kn = torch.randn(B, max(Sn, 1), Hkv, D, generator=g).to(dt)[:, :Sn]
    vn = torch.randn(B, max(Sn, 1), Hkv, D, generator=g).to(dt)[:, :Sn]
    return kc, vc, q, kn, vn


def cache_flat_statement(key, value, k_rows, v_rows):
    """cache_kernels.cu:483-520 as flat index arithmetic on the raw storage: token t, element i -> cache[t * cache_stride + i].
    key / value [n, heads, d] contiguous; k_rows / v_rows: 2-D numpy VIEWS [rows, heads * d] of the slot's row-block from the
    append position on (their row stride is the kernel's k_cache_stride)."""
    n, width = key.shape[0], key.shape[1] * key.shape[2]
    kf, vf = key.reshape(-1), value.reshape(-1)
    key_stride = value_stride = width                          # cache.cpp:40-46 passes key.stride(0)
    for t in range(n):                                         # blockIdx.x
        for i in range(width):                                 # threadIdx.x strided loop
            k_rows[t, i] = kf[t * key_stride + i]
            v_rows[t, i] = vf[t * value_stride + i]


def operator_by_composition(ref, case, seed):
    name, B, Sq, Sn, lens, slots, idx, Hq, Hkv, D, causal, dtype = case
    kc, vc, q, kn, vn = op_inputs(seed, B, Sq, Sn, lens, slots, Hq, Hkv, D, dtype)
    # work on int16 views of the storage: the append is a copy of bit patterns, whatever the dtype
    kci, vci = kc.view(torch.int16).numpy(), vc.view(torch.int16).numpy()
    G = Hq // Hkv
    outs, lses, masked = [], [], []
    for b in range(B):
        slot = idx[b] if idx is not None else b
        if Sn:
            cache_flat_statement(kn[b].contiguous().view(torch.int16).numpy(), vn[b].contiguous().view(torch.int16).numpy(),
                                 kci[slot, lens[b]:].reshape(-1, Hkv * D), vci[slot, lens[b]:].reshape(-1, Hkv * D))
    for b in range(B):
        slot = idx[b] if idx is not None else b
        Lk = lens[b] + Sn
        kk = kc[slot:slot + 1, :Lk].repeat_interleave(G, dim=2)
        vv = vc[slot:slot + 1, :Lk].repeat_interleave(G, dim=2)
        mask = None
        if causal:
            mask = torch.triu(torch.full([1, Sq, Lk], float("-inf"), dtype=torch.float32), diagonal=1 + Lk - Sq)
        o, lse = ref(q[b:b + 1], kk, vv, mask)
        dead = torch.zeros(Sq, dtype=torch.bool)
        if causal and Sq > Lk:
            dead[:Sq - Lk] = True                              # these rows see no key: softmax of an empty set
        o = o.float()
        o[0, dead] = 0.0                                       # the operator's convention (flash_attn_interface.py: fully masked rows -> 0)
        outs.append(o)
        lses.append(lse.float())
        masked.append(dead)
    return torch.cat(outs), torch.cat(lses), torch.stack(masked), kc, vc


def main():
    ref = reference_functions()
    out = {}
    for i, case in enumerate(OP_CASES):
        seed = 5000 + i
        o, lse, dead, kc, vc = operator_by_composition(ref, case, seed)
        name = "op_" + case[0]
        out[name + "/out"] = o.numpy()
        out[name + "/lse"] = lse.numpy()
        out[name + "/masked_rows"] = dead.numpy()
        # the caches after the append, as a checksum per slot (the appended rows are additionally compared bit for bit in the test,
        # which re-creates the inputs from the seed)
        out[name + "/k_sum"] = kc.view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)).numpy()
        out[name + "/v_sum"] = vc.view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)).numpy()
        out[name + "/seed"] = np.array([seed], dtype=np.int64)
    for i, (name, B, Sq, Sk, Hq, Hkv, D, causal, dtype) in enumerate(CASES):
        seed = 1000 + i
        q, k, v = inputs(seed, B, Sq, Sk, Hq, Hkv, D, dtype)
        # the reference function has no GQA: hand it each query head's kv head (h -> h // (Hq / Hkv), flash_attn_interface.py:1180-1184)
        kk = k.repeat_interleave(Hq // Hkv, dim=2)
        vv = v.repeat_interleave(Hq // Hkv, dim=2)
        mask = None
        if causal:
            # row i sees keys j <= i + (Sk - Sq); for Sq == Sk this IS the example's torch.triu(-inf, diagonal=1)
            mask = torch.triu(torch.full([1, Sq, Sk], float("-inf"), dtype=torch.float32), diagonal=1 + Sk - Sq)
        o, lse = ref(q, kk, vv, mask)
        out[name + "/out"] = o.float().numpy()
        out[name + "/lse"] = lse.float().numpy()            # [B, Hq, Sq], natural log, of the scaled scores
        out[name + "/meta"] = np.array([seed, B, Sq, Sk, Hq, Hkv, D, int(causal), {"float16": 0, "bfloat16": 1}[dtype]], dtype=np.int64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(CASES), "+", len(OP_CASES), "cases")


if __name__ == "__main__":
    main()
