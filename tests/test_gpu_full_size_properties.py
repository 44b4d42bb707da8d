"""Size-independent properties of the operator at BASELINE.json's FULL sizes (configs[1]: Yi-6B heads, 32 702-token prompt,
batch-16 decode at 32 k; configs[3]: 128 k context) — they cover EVERY output element of the full-size launches, complementing
the direct oracle comparison of tests/test_gpu_full_size_parity.py (full decode batches, sampled query blocks of the prefills).
  * rows of softmax sum to one:           V = 1            ->  O = 1
  * linearity in V:                        O(V1 + V2)       =  O(V1) + O(V2)
  * chunked prefill == whole-prompt prefill (the reference's Sarathi vs vLLM scheduling of the same request)
  * split-KV invariance:                   any split count gives the same decode / prefill result
  * key-order invariance of decode:        permuting the cached (k, v) rows together leaves the output unchanged
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _fa():
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    return flash_attn_with_kvcache


def test_c2_whole_prompt_rows_sum_to_one_and_linearity():
    fa = _fa()
    torch.manual_seed(0)
    n, Hq, Hkv = 32702, 32, 4
    q = torch.randn(1, n, Hq, D, device=DEV).half()
    k = torch.randn(1, n, Hkv, D, device=DEV).half()
    ones = torch.ones(1, n, Hkv, D, device=DEV).half()
    cl = torch.tensor([n], dtype=torch.int32, device=DEV)
    o = fa(q, k, ones, cache_seqlens=cl, causal=True)
    assert (o.float() - 1.0).abs().max().item() <= 2e-3
    v1 = torch.randn(1, n, Hkv, D, device=DEV).half()
    v2 = torch.randn(1, n, Hkv, D, device=DEV).half()
    o1, o2 = fa(q, k, v1, cache_seqlens=cl, causal=True), fa(q, k, v2, cache_seqlens=cl, causal=True)
    o12 = fa(q, k, (v1.float() + v2.float()).half(), cache_seqlens=cl, causal=True)
    # v1 + v2 is rounded to f16 once more than the separate runs: 2^-11 relative on |v| ~ 3
    assert (o12.float() - (o1.float() + o2.float())).abs().max().item() <= 6e-3


def test_c2_chunked_prefill_equals_whole_prompt():
    fa = _fa()
    torch.manual_seed(1)
    n, Hq, Hkv, chunk = 32702, 32, 4, 4096
    q = torch.randn(1, n, Hq, D, device=DEV).half()
    k = torch.randn(1, n, Hkv, D, device=DEV).half()
    v = torch.randn(1, n, Hkv, D, device=DEV).half()
    whole = fa(q, k, v, cache_seqlens=torch.tensor([n], dtype=torch.int32, device=DEV), causal=True)
    worst = 0.0
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        part = fa(q[:, s0:s0 + m], k, v, cache_seqlens=torch.tensor([s0 + m], dtype=torch.int32, device=DEV), causal=True,
                  _max_seqlen_k=s0 + m)
        worst = max(worst, (part.float() - whole[:, s0:s0 + m].float()).abs().max().item())
    assert worst <= 2e-3, worst


def test_c2_decode_batch16_at_32k_properties():
    fa = _fa()
    torch.manual_seed(2)
    B, ctx, Hq, Hkv = 16, 32768, 32, 4
    q = torch.randn(B, 1, Hq, D, device=DEV).half()
    k = torch.randn(B, ctx, Hkv, D, device=DEV).half()
    v = torch.randn(B, ctx, Hkv, D, device=DEV).half()
    cl = torch.randint(ctx // 2, ctx, (B,), dtype=torch.int32, device=DEV)
    ones = torch.ones_like(v)
    assert (fa(q, k, ones, cache_seqlens=cl, causal=True).float() - 1.0).abs().max().item() <= 2e-3
    base = fa(q, k, v, cache_seqlens=cl, causal=True, num_splits=1)
    for splits in (0, 5, 12, 48):
        o = fa(q, k, v, cache_seqlens=cl, causal=True, num_splits=splits)
        assert (o.float() - base.float()).abs().max().item() <= 1e-3, splits
    # permute the visible rows of every sequence (keys and values together)
    kp, vp = k.clone(), v.clone()
    for b in range(B):
        L = int(cl[b])
        perm = torch.randperm(L, device=DEV)
        kp[b, :L] = k[b, :L][perm]
        vp[b, :L] = v[b, :L][perm]
    o = fa(q, kp, vp, cache_seqlens=cl, causal=True)
    assert (o.float() - base.float()).abs().max().item() <= 1e-3


def test_c4_128k_context_rows_sum_to_one():
    """Yi-34B/TP2 rank shape at 131 072 tokens: 16 k chunk on a 112 k prefix, and a batch-8 decode."""
    fa = _fa()
    torch.manual_seed(3)
    Hq, Hkv, ctx, n = 28, 4, 131072, 16384
    k = torch.randn(1, ctx, Hkv, D, device=DEV).half()
    ones = torch.ones(1, ctx, Hkv, D, device=DEV).half()
    q = torch.randn(1, n, Hq, D, device=DEV).half()
    o = fa(q, k, ones, cache_seqlens=torch.tensor([ctx], dtype=torch.int32, device=DEV), causal=True, _max_seqlen_k=ctx)
    assert (o.float() - 1.0).abs().max().item() <= 2e-3
    qd = torch.randn(8, 1, Hq, D, device=DEV).half()
    od = fa(qd, k.expand(8, -1, -1, -1), ones.expand(8, -1, -1, -1), cache_seqlens=torch.full((8,), ctx, dtype=torch.int32, device=DEV), causal=True)
    assert (od.float() - 1.0).abs().max().item() <= 2e-3
