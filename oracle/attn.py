"""CPU oracle for the attention half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED (stated as the task requires): the arithmetic of this path lives in the
third-party package  flash-attn == 2.5.9.post1  (/root/reference/sarathi-lean/requirements.txt:22),
whose sources are NOT under /root/reference and which cannot be built or imported here (CUDA
only, no network).  The reference repository holds no golden vectors and no unit test for it
(SURVEY §0.3, §8c).  This file therefore restates the *published* algorithm, following the
in-tree FlashAttention-2.6.1 fork that documents the same operator:

    semantics / docstring   /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291
    argument rules          /root/reference/pod_attn/pod_attn/flash_api.cpp:1291-1578
    visible keys            /root/reference/pod_attn/pod_attn/block_info.h:22-23
    causal mask             /root/reference/pod_attn/pod_attn/mask.h:164-196  (bottom-right aligned)
    softmax numerics        /root/reference/pod_attn/pod_attn/softmax.h:69-157
    KV append               /root/reference/sarathi-lean/csrc/cache_kernels.cu:482-570 (cache_flat)

and anchors on the reference's own call sites
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:151-205).
It is cross-checked against torch's independent CPU scaled_dot_product_attention and — the one pure-PyTorch
attention statement that does live under /root/reference — against outputs + LSEs of `ref_mha_bmhk` of the vendored
CUTLASS example (pod_attn/csrc/cutlass/examples/41_fused_multi_head_attention/fmha_backward_test.py:78-105), run here
from the reference's own file by oracle/gen_golden_attn_intree.py and committed as tests/golden/attn_intree_ref_mha.npz
(tests/test_attn_oracle.py).  That function is not the operator's test (it has no KV cache, no GQA, no bottom-right
mask of its own): it pins softmax(q.k^T/sqrt(d) + mask).v and the LSE, not the operator's semantics — hence still
"unpinned" in the task's sense.

What pins what (rounds 4 and 6 — "partial" made as tight as the tree allows; there is still no fixture of the OPERATOR itself):

    semantic of flash_attn_with_kvcache            pinned by                                                          how
    ---------------------------------------------  -----------------------------------------------------------------  ---------------------------
    softmax(q.k^T.scale + mask).v, fp32, and LSE   tests/golden/attn_intree_ref_mha.npz, 5 plain + 20 composed cases   in-tree ref_mha_bmhk, executed
    append position and content (k, v given)       op_* cases: caches after the call == after the reference's          cache_kernels.cu:483-520 restated
                                                   cache_flat statement at row cache_seqlens[b] (checksums + rows)     as flat index arithmetic
    cache_batch_idx (slot of batch entry b)        op_* cases with permuted / partial slot lists                       composition (row selection)
    cache_seqlens (+ seqlen_new) visible keys      op_* cases, Lk in {1, 5, 63, 64, 65, 300, 4097}                     composition (cut)
    strided [:, :max_len] cache views              op_strided_view                                                     composition
    bottom-right causal alignment                  the docstring's literal 2 x 5 AND 5 x 2 keep / mask matrices        DOCSTRING flash_attn_interface.py:1192-1202,
                                                   read back from the operator's output (tests/test_docstring_pins.py,  asserted on this file and on the HIP
                                                   tests/test_gpu_docstring_pins.py); op_* chunk cases                  kernels (round 6); mask.h:164-196
    GQA head mapping h -> h // (Hq / Hkv)          the docstring's "6 heads over 2: heads 0, 1, 2 -> kv head 0,         DOCSTRING :1187-1190, asserted on this file
                                                   3, 4, 5 -> kv head 1" (same two test files); op_* groups 1-8         and on the HIP kernels (round 6)
    rows that see no key -> 0                      the docstring's ":1203 If the row of the mask is all zero, the       DOCSTRING :1203 (output); LSE = +inf for
                                                   output will be zero" on the 5 x 2 example; op_chunk_longer_than_keys  such a row stays OURS (ref_mha_bmhk: NaN)
    fp16 agreement at atol = 1e-3                  tests/test_gpu_attention.py::test_pod_sweep_shapes_...               the reference's own GPU-vs-GPU
                                                   (POD sweep shapes, pod_attn/tests/attn_sweep.py:82-97)              criterion, kernels vs this file

The HIP kernels are additionally compared with the composed vectors DIRECTLY (test_kernels_against_the_operator_by_composition_vectors).

Two precisions:
  * ``math="f64"``  — exact-arithmetic ground truth on the fp16/bf16 inputs (what tests compare to);
  * ``math="f32"``  — fp32 accumulate, P rounded to the I/O dtype before PV (the reference kernel's
                      numerics, flash_fwd_kernel.h:927,992) — used to bound the allowed error and as
                      the timed CPU baseline.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

APPEND_ERR = "If key is supplied, it must have seqlen <= the seqlen of the KV cache"   # flash_api.cpp:1454


def cache_flat_ref(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                   kv_cache_dtype: str = "auto") -> None:
    """cache_kernels.cu:482-570: k_cache[t] = key[t], v_cache[t] = value[t] for t < num_tokens.
    key/value [n, kvh, D]; caches are row views [>=n, kvh, D] (caller pre-slices at the offset)."""
    if kv_cache_dtype != "auto":
        raise RuntimeError("Unsupported data type of kv cache: " + kv_cache_dtype)   # :532-534
    assert k_cache.stride(0) == v_cache.stride(0)                                     # :543
    n = key.shape[0]
    k_cache[:n].copy_(key)
    v_cache[:n].copy_(value)


def flash_attn_with_kvcache_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                                k: Optional[torch.Tensor] = None, v: Optional[torch.Tensor] = None,
                                cache_seqlens: Optional[Union[int, torch.Tensor]] = None,
                                cache_batch_idx: Optional[torch.Tensor] = None,
                                softmax_scale: Optional[float] = None, causal: bool = False,
                                math: str = "f64", return_lse: bool = False):
    """q [B,Sq,Hq,D]; caches [Bc,Sk,Hkv,D]; optional new k,v [B,Sn,Hkv,D] appended in place at row
    cache_seqlens[b] of cache slot cache_batch_idx[b] (identity if None).  Returns [B,Sq,Hq,D] in
    float64 (math="f64") or the input dtype (math="f32")."""
    B, Sq, Hq, D = q.shape
    Bc, Sk, Hkv, _ = k_cache.shape
    assert Hq % Hkv == 0
    G = Hq // Hkv
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    if cache_seqlens is None:
        lens = [Sk] * B                                   # no seqlens -> whole cache visible
    elif isinstance(cache_seqlens, int):
        lens = [cache_seqlens] * B
    else:
        lens = [int(x) for x in cache_seqlens.tolist()]
    idx = list(range(B)) if cache_batch_idx is None else [int(x) for x in cache_batch_idx.tolist()]
    Sn = 0
    if k is not None:
        assert v is not None and cache_seqlens is not None       # flash_api.cpp:1452-1453
        Sn = k.shape[1]
        for b in range(B):                                        # append-then-attend
            k_cache[idx[b], lens[b]:lens[b] + Sn] = k[b]
            v_cache[idx[b], lens[b]:lens[b] + Sn] = v[b]
    wt = torch.float64 if math == "f64" else torch.float32
    out = torch.zeros(B, Sq, Hq, D, dtype=wt)
    lse = torch.full((B, Hq, Sq), float("-inf"), dtype=wt)
    for b in range(B):
        Lk = lens[b] + Sn                                          # block_info.h:22-23
        if Lk <= 0:
            continue
        # GQA: query head h uses kv head h // G (flash_attn_interface.py:1189-1192).  The G query heads of a kv head are
        # stacked along the row axis instead of repeating K/V G times (same arithmetic, 1/G of the memory).
        Kh = k_cache[idx[b], :Lk].to(wt).permute(1, 0, 2)          # [Hkv,Lk,D]
        Vh = v_cache[idx[b], :Lk].to(wt).permute(1, 0, 2)
        QB = max(8, min(512, (1 << 27) // max(1, Hq * Lk)))         # query rows per block (bounds memory only)
        for q0 in range(0, Sq, QB):
            q1 = min(Sq, q0 + QB)
            sq = q1 - q0
            Qb = q[b, q0:q1].to(wt).reshape(sq, Hkv, G, D).permute(1, 2, 0, 3).reshape(Hkv, G * sq, D)
            S = torch.matmul(Qb, Kh.transpose(1, 2)) * softmax_scale   # [Hkv,G*sq,Lk]
            if causal and Sq > 1:                                   # Sq==1: causal dropped (flash_api.cpp:1364)
                i = torch.arange(q0, q1).repeat(G).view(-1, 1)      # row r of the stack is query q0 + r % sq
                j = torch.arange(Lk).view(1, Lk)
                S = S.masked_fill(~(j <= i + (Lk - Sq)), float("-inf"))  # bottom-right aligned (mask.h:164-196)
            m = S.max(dim=-1, keepdim=True).values
            dead = torch.isinf(m) & (m < 0)                        # fully masked row -> output 0
            m = torch.where(dead, torch.zeros_like(m), m)
            P = torch.exp(S - m)
            del S
            l = P.sum(dim=-1, keepdim=True)
            if math == "f32":
                P = P.to(q.dtype).to(wt)                           # P rounded before PV
            O = torch.matmul(P, Vh) / torch.where(dead, torch.ones_like(l), l)
            del P
            O = torch.where(dead, torch.zeros_like(O), O)          # [Hkv,G*sq,D]
            out[b, q0:q1] = O.view(Hkv, G, sq, D).permute(2, 0, 1, 3).reshape(sq, Hq, D)
            row_lse = (m + torch.log(l)).squeeze(-1)               # [Hkv,G*sq]
            row_lse = torch.where(dead.squeeze(-1), torch.full_like(row_lse, float("inf")), row_lse)
            lse[b, :, q0:q1] = row_lse.view(Hq, sq)
    if math == "f32":
        out = out.to(q.dtype)
    return (out, lse) if return_lse else out


def flash_attn_func_ref(q, k, v, softmax_scale=None, causal=False, math="f64"):
    """flash_attn_func (no cache): q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D]."""
    return flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=k.shape[1],
                                       softmax_scale=softmax_scale, causal=causal, math=math)


def check_append_shapes(seqlen_q_effective: int, seqlen_k_cache: int) -> None:
    """flash_api.cpp:1454 — the reference raises when (possibly GQA-swapped) seqlen_q exceeds the
    cache's seqlen dimension.  SURVEY §A.2: the product's shim does not raise there."""
    if seqlen_q_effective > seqlen_k_cache:
        raise RuntimeError(APPEND_ERR)


# ---------------------------------------------------------------------------------------------------------------------
# RoPE: the reference applies it to q and k BEFORE the attention wrapper
# (sarathi/model_executor/models/yi.py:172-173 -> layers/rotary_embedding.py:86-101 -> csrc/pos_encoding_kernels.cu)
# ---------------------------------------------------------------------------------------------------------------------
def make_cos_sin_cache(rotary_dim: int, max_position: int, base: float = 10000.0, dtype=torch.float16) -> torch.Tensor:
    """rotary_embedding.py:55-84: [max_position, rotary_dim] = cat(cos, sin) of t x inv_freq, computed in float and cast to the
    model dtype (the cache is an INPUT of the rotary kernel; tests hand the same cache to the oracle and to the GPU path)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)


def rotary_embedding_ref(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                         cos_sin_cache: torch.Tensor, is_neox: bool = True) -> None:
    """pos_encoding_kernels.cu:9-77, in place on query [T, Hq*hs] and key [T, Hkv*hs].  NeoX style pairs element i with element
    i + rot_dim/2 (:18-23), GPT-J style pairs 2i with 2i+1 (:24-30).  The arithmetic is carried in the tensors' own dtype
    (`scalar_t`, :32-35): every product and the final sum are rounded to it — restated as float ops each rounded to the dtype."""
    dt = query.dtype
    rot_dim = cos_sin_cache.shape[1]
    e = rot_dim // 2
    cs = cos_sin_cache[positions.long()]                       # [T, rot_dim]
    cos, sin = cs[:, :e].float().unsqueeze(1), cs[:, e:].float().unsqueeze(1)      # [T, 1, e]
    rnd = lambda x: x.to(dt).float()
    for arr in (query, key):
        T = arr.shape[0]
        a = arr.view(T, -1, head_size)
        if is_neox:
            x, y = a[..., :e].float(), a[..., e:rot_dim].float()
        else:
            x, y = a[..., 0:rot_dim:2].float(), a[..., 1:rot_dim:2].float()
        nx = rnd(rnd(x * cos) - rnd(y * sin))
        ny = rnd(rnd(y * cos) + rnd(x * sin))
        if is_neox:
            a[..., :e] = nx.to(dt)
            a[..., e:rot_dim] = ny.to(dt)
        else:
            a[..., 0:rot_dim:2] = nx.to(dt)
            a[..., 1:rot_dim:2] = ny.to(dt)
