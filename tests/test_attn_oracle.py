"""oracle/attn.py against an independent implementation (torch CPU SDPA) and its own invariants."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle.attn import cache_flat_ref, flash_attn_func_ref, flash_attn_with_kvcache_ref


def _sdpa(q, k, v, mask):
    # q [B,Sq,Hq,D] k/v [B,Sk,Hkv,D]
    G = q.shape[2] // k.shape[2]
    kk = k.double().permute(0, 2, 1, 3).repeat_interleave(G, dim=1)
    vv = v.double().permute(0, 2, 1, 3).repeat_interleave(G, dim=1)
    o = F.scaled_dot_product_attention(q.double().permute(0, 2, 1, 3), kk, vv, attn_mask=mask)
    return o.permute(0, 2, 1, 3)


@pytest.mark.parametrize("Sq,Sk,Hq,Hkv", [(1, 77, 8, 2), (33, 33, 4, 4), (17, 90, 8, 1), (64, 300, 6, 2)])
def test_matches_sdpa(Sq, Sk, Hq, Hkv):
    torch.manual_seed(0)
    B, D = 2, 64
    q = torch.randn(B, Sq, Hq, D).half()
    k = torch.randn(B, Sk, Hkv, D).half()
    v = torch.randn(B, Sk, Hkv, D).half()
    i = torch.arange(Sq).view(Sq, 1)
    j = torch.arange(Sk).view(1, Sk)
    mask = j <= i + (Sk - Sq)
    ref = _sdpa(q, k, v, mask)
    got = flash_attn_func_ref(q, k, v, causal=True)
    assert torch.allclose(got, ref, atol=1e-9, rtol=1e-9)
    got_nc = flash_attn_func_ref(q, k, v, causal=False)
    assert torch.allclose(got_nc, _sdpa(q, k, v, None), atol=1e-9, rtol=1e-9)


def test_bottom_right_alignment_docstring_examples():
    # flash_attn_interface.py:1194-1206: seqlen_q=2,seqlen_k=5 and seqlen_q=5,seqlen_k=2
    torch.manual_seed(1)
    q = torch.randn(1, 5, 1, 16).half()
    k = torch.randn(1, 2, 1, 16).half()
    v = torch.randn(1, 2, 1, 16).half()
    o = flash_attn_func_ref(q, k, v, causal=True)
    assert torch.all(o[0, :3] == 0)                     # fully masked rows -> 0
    assert torch.allclose(o[0, 3, 0], v[0, 0, 0].double())   # row 3 sees only key 0
    assert not torch.all(o[0, 4] == 0)


def test_append_and_batch_idx_and_chunked_equals_unchunked():
    torch.manual_seed(2)
    Hq, Hkv, D, ctx = 8, 2, 64, 256
    kc = torch.zeros(5, ctx, Hkv, D).half()
    vc = torch.zeros(5, ctx, Hkv, D).half()
    n = 100
    q = torch.randn(1, n, Hq, D).half()
    k = torch.randn(n, Hkv, D).half()
    v = torch.randn(n, Hkv, D).half()
    slot = 3
    # unchunked prefill through cache_flat + kvcache attention (wrapper :151-166)
    cache_flat_ref(k, v, kc[slot], vc[slot])
    full = flash_attn_with_kvcache_ref(q, kc[slot:slot + 1], vc[slot:slot + 1], cache_seqlens=torch.tensor([n], dtype=torch.int32), causal=True)
    # chunked: 4 x 25
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    outs = []
    for c in range(0, n, 25):
        cache_flat_ref(k[c:c + 25], v[c:c + 25], kc2[slot][c:], vc2[slot][c:])
        outs.append(flash_attn_with_kvcache_ref(q[:, c:c + 25], kc2[slot:slot + 1], vc2[slot:slot + 1],
                                                cache_seqlens=torch.tensor([c + 25], dtype=torch.int32), causal=True))
    assert torch.allclose(torch.cat(outs, 1), full, atol=1e-12)
    # decode with append + cache_batch_idx
    qd = torch.randn(2, 1, Hq, D).half()
    kn = torch.randn(2, 1, Hkv, D).half()
    vn = torch.randn(2, 1, Hkv, D).half()
    cache_flat_ref(k[:40], v[:40], kc[1], vc[1])
    lens = torch.tensor([n, 40], dtype=torch.int32)
    idx = torch.tensor([slot, 1], dtype=torch.int32)
    o = flash_attn_with_kvcache_ref(qd, kc[:, :n + 1], vc[:, :n + 1], kn, vn, cache_seqlens=lens, cache_batch_idx=idx, causal=True)
    assert torch.equal(kc[slot, n], kn[0, 0]) and torch.equal(vc[1, 40], vn[1, 0])
    ref0 = flash_attn_func_ref(qd[:1], kc[slot:slot + 1, :n + 1], vc[slot:slot + 1, :n + 1])
    assert torch.allclose(o[:1], ref0, atol=1e-12)


def test_f32_mode_error_budget():
    torch.manual_seed(3)
    q = torch.randn(1, 64, 8, 128).half()
    k = torch.randn(1, 512, 2, 128).half()
    v = torch.randn(1, 512, 2, 128).half()
    a = flash_attn_func_ref(q, k, v, causal=True)
    b = flash_attn_func_ref(q, k, v, causal=True, math="f32")
    assert b.dtype == torch.float16
    assert (a - b.double()).abs().max() < 2e-3


def test_cache_flat_rejects_dtype():
    k = torch.zeros(2, 1, 8).half()
    with pytest.raises(RuntimeError, match="Unsupported data type"):
        cache_flat_ref(k, k, k.clone(), k.clone(), "fp8")


def test_against_a_scalar_python_restatement():
    """Third, fully independent leg for the oracle: a scalar pure-Python evaluation of the published operator
    (flash_attn_interface.py:1168-1254: append at cache_seqlens, GQA head mapping h -> h // (Hq/Hkv), bottom-right aligned
    causal mask, rows with no visible key -> 0) on tiny shapes, including seqlen_q > seqlen_k and cache_batch_idx."""
    import math
    torch.manual_seed(3)
    B, Sq, Hq, Hkv, D, Sk = 2, 5, 4, 2, 8, 9
    q = torch.randn(B, Sq, Hq, D, dtype=torch.float64)
    kc = torch.randn(3, Sk, Hkv, D, dtype=torch.float64)
    vc = torch.randn(3, Sk, Hkv, D, dtype=torch.float64)
    kn = torch.randn(B, 2, Hkv, D, dtype=torch.float64)
    vn = torch.randn(B, 2, Hkv, D, dtype=torch.float64)
    cls = [1, 6]                      # entry 0: 1 cached + 2 new = 3 keys < 5 queries -> the first two query rows see nothing
    idx = [2, 0]
    for causal in (True, False):
        k2, v2 = kc.clone(), vc.clone()
        got = flash_attn_with_kvcache_ref(q, k2, v2, kn, vn, cache_seqlens=torch.tensor(cls, dtype=torch.int32),
                                          cache_batch_idx=torch.tensor(idx, dtype=torch.int32), causal=causal)
        ke, ve = kc.clone(), vc.clone()
        scale = 1.0 / math.sqrt(D)
        for b in range(B):
            slot, c = idx[b], cls[b]
            for t in range(2):
                ke[slot, c + t] = kn[b, t]
                ve[slot, c + t] = vn[b, t]
            Lk = c + 2
            for i in range(Sq):
                for h in range(Hq):
                    hk = h // (Hq // Hkv)
                    vis = [j for j in range(Lk) if (not causal) or j <= i + (Lk - Sq)]
                    if not vis:
                        want = [0.0] * D
                    else:
                        s = [scale * sum(float(q[b, i, h, d]) * float(ke[slot, j, hk, d]) for d in range(D)) for j in vis]
                        m = max(s)
                        w = [math.exp(x - m) for x in s]
                        z = sum(w)
                        want = [sum(w[n] * float(ve[slot, j, hk, d]) for n, j in enumerate(vis)) / z for d in range(D)]
                    for d in range(D):
                        assert abs(float(got[b, i, h, d]) - want[d]) < 1e-9, (causal, b, i, h, d)
        assert torch.equal(k2, ke) and torch.equal(v2, ve)       # the append itself


def test_rotary_oracle_matches_scalar_half_arithmetic_and_is_a_rotation():
    """oracle/attn.py rotary_embedding_ref vs a scalar evaluation of pos_encoding_kernels.cu:9-37 in torch half scalars (each
    operator rounds to half), NeoX and GPT-J pairings; position 0 is the identity; norms are preserved to rounding."""
    from oracle.attn import make_cos_sin_cache, rotary_embedding_ref
    torch.manual_seed(0)
    hs, T = 64, 9
    cs = make_cos_sin_cache(hs, 256)
    pos = torch.tensor([0, 1, 2, 17, 100, 255, 3, 3, 8])
    for neox in (True, False):
        q, k = torch.randn(T, 3 * hs).half(), torch.randn(T, 2 * hs).half()
        q0, k0 = q.clone(), k.clone()
        rotary_embedding_ref(pos, q, k, hs, cs, neox)
        e = hs // 2
        for arr, arr0, nh in ((q, q0, 3), (k, k0, 2)):
            for t in range(T):
                for h in range(nh):
                    for i in (0, 1, 13, e - 1):
                        xi, yi = (i, e + i) if neox else (2 * i, 2 * i + 1)
                        x, y = arr0[t, h * hs + xi], arr0[t, h * hs + yi]
                        c, s = cs[pos[t], i], cs[pos[t], e + i]
                        assert arr[t, h * hs + xi] == x * c - y * s and arr[t, h * hs + yi] == y * c + x * s
        assert torch.equal(q[0], q0[0])
        assert torch.allclose(q.float().view(T, 3, hs).norm(dim=-1), q0.float().view(T, 3, hs).norm(dim=-1), rtol=3e-3)


def _intree_cases():
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attn_intree_ref_mha.npz")
    z = np.load(path)
    return z, sorted({k.split("/")[0] for k in z.files})


def _op_case(name):
    """Inputs of an operator-by-composition case, re-created from its seed with the generating script's own recipe."""
    import oracle.gen_golden_attn_intree as gen
    z, _ = _intree_cases()
    case = next(c for c in gen.OP_CASES if "op_" + c[0] == name)
    _n, B, Sq, Sn, lens, slots, idx, Hq, Hkv, D, causal, dtype = case
    kc, vc, q, kn, vn = gen.op_inputs(int(z[name + "/seed"][0]), B, Sq, Sn, lens, slots, Hq, Hkv, D, dtype)
    return z, case, kc, vc, q, kn, vn


@pytest.mark.parametrize("name", [n for n in _intree_cases()[1] if n.startswith("op_")])
def test_operator_by_composition_vectors(name):
    """The OPERATOR's index semantics against vectors composed from reference text (oracle/gen_golden_attn_intree.py, OP_CASES): the
    append executed as the reference's cache_flat statement, the slot picked by cache_batch_idx, the keys cut at cache_seqlens (+ new
    tokens), attention by the in-tree ref_mha_bmhk.  The oracle must produce the same outputs and LSEs from ONE call of the operator
    it restates — and leave the caches in the state the cache_flat statement left them."""
    z, case, kc, vc, q, kn, vn = _op_case(name)
    _n, B, Sq, Sn, lens, slots, idx, Hq, Hkv, D, causal, dtype = case
    cl = torch.tensor(lens, dtype=torch.int32)
    bi = torch.tensor(idx, dtype=torch.int32) if idx is not None else None
    ml = max(lens) + Sn
    kview, vview = (kc[:, :ml], vc[:, :ml]) if "strided" in name else (kc, vc)
    got, lse = flash_attn_with_kvcache_ref(q, kview, vview, kn if Sn else None, vn if Sn else None, cache_seqlens=cl, cache_batch_idx=bi,
                                           causal=bool(causal), math="f64", return_lse=True)
    ref = torch.from_numpy(z[name + "/out"]).double()
    dead = torch.from_numpy(z[name + "/masked_rows"])
    assert torch.allclose(got, ref, atol=3e-5, rtol=3e-5), (got - ref).abs().max().item()
    if dead.any():
        assert float(got[dead].abs().max()) == 0.0                 # rows that see no key
    ref_lse = torch.from_numpy(z[name + "/lse"]).double()           # [B, Hq, Sq]
    live = ~dead[:, None, :].expand(-1, Hq, -1)
    assert torch.allclose(lse[live], ref_lse[live], atol=3e-5, rtol=3e-6)
    # the caches: checksums of every slot as the cache_flat statement left them, and the appended rows bit for bit
    assert torch.equal(kc.view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)), torch.from_numpy(z[name + "/k_sum"]))
    assert torch.equal(vc.view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)), torch.from_numpy(z[name + "/v_sum"]))
    for b in range(B):
        slot = idx[b] if idx is not None else b
        if Sn:
            assert torch.equal(kc[slot, lens[b]:lens[b] + Sn], kn[b]) and torch.equal(vc[slot, lens[b]:lens[b] + Sn], vn[b])


@pytest.mark.parametrize("name", [n for n in _intree_cases()[1] if not n.startswith("op_")])
def test_matches_the_in_tree_pytorch_mha_reference_vectors(name):
    """tests/golden/attn_intree_ref_mha.npz: outputs and LSEs of `ref_mha_bmhk` of the CUTLASS example vendored inside the reference
    tree (…/41_fused_multi_head_attention/fmha_backward_test.py:78-105), produced in this container by
    oracle/gen_golden_attn_intree.py from the reference's own file.  fp32 arithmetic on the fp16 / bf16 inputs there, fp64 here."""
    z, _ = _intree_cases()
    seed, B, Sq, Sk, Hq, Hkv, D, causal, dt = [int(x) for x in z[name + "/meta"]]
    g = torch.Generator().manual_seed(seed)
    dtype = torch.bfloat16 if dt else torch.float16
    q = torch.randn(B, Sq, Hq, D, generator=g).to(dtype)          # the generating script's recipe (inputs())
    k = torch.randn(B, Sk, Hkv, D, generator=g).to(dtype)
    v = torch.randn(B, Sk, Hkv, D, generator=g).to(dtype)
    got, lse = flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=torch.full((B,), Sk, dtype=torch.int32), causal=bool(causal),
                                           math="f64", return_lse=True)
    ref = torch.from_numpy(z[name + "/out"]).double()
    assert torch.allclose(got, ref, atol=2e-5, rtol=2e-5), (got - ref).abs().max().item()
    ref_lse = torch.from_numpy(z[name + "/lse"]).double()            # [B, Hq, Sq], natural log of sum exp(scaled scores)
    assert torch.allclose(lse, ref_lse, atol=2e-5, rtol=2e-6), (lse - ref_lse).abs().max().item()
    # and the reference-numerics mode (fp32 accumulate, P rounded to the I/O dtype) stays within the I/O dtype's rounding of them
    got32 = flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=torch.full((B,), Sk, dtype=torch.int32), causal=bool(causal), math="f32")
    tol = 8e-3 if dt else 2e-3
    assert torch.allclose(got32.double(), ref, atol=tol, rtol=tol)
