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


def main():
    ref = reference_functions()
    out = {}
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
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(CASES), "cases")


if __name__ == "__main__":
    main()
