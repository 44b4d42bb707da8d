"""Parity AT BASELINE.json's full sizes against the CPU oracle (not only through properties):
  configs[1] (c2, Yi-6B TP1):   batch-16 decode at 32 k in full; the 32 702-token whole-prompt prefill on sampled query blocks
  configs[3] (c4, Yi-34B TP2):  batch-8 decode at 128 k in full; a 16 k chunk on a 112 k prefix on sampled query blocks
The oracle of query rows [a, b) of a causal prefill over Lk keys is exactly the chunked call q[:, a:b] with
cache_seqlens = (Lk - Sq) + b (bottom-right alignment, mask.h:164-196), so a 256-row block x all heads costs seconds.
Tolerances are those of tests/test_gpu_attention.py (fp16: atol = rtol = 2e-3 vs float64, and max error within 2x the error of
the reference-numerics CPU run + 1e-5).
"""
import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _check(got, ref64, ref32, what, atol=2e-3, rtol=2e-3):
    err = (got.double().cpu() - ref64).abs()
    bound = atol + rtol * ref64.abs()
    assert bool((err <= bound).all()), "%s: max err %.3e (allowed %.3e)" % (what, err.max().item(), bound.max().item())
    e_ref = (ref32.double() - ref64).abs().max().item()
    assert err.max().item() <= 2 * e_ref + 1e-5, "%s: kernel err %.3e vs reference-numerics err %.3e" % (what, err.max().item(), e_ref)


def _decode_case(B, ctx, Hq, Hkv, seed):
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(seed)
    slots = B + 2
    kc = torch.randn(slots, ctx, Hkv, D).half()
    vc = torch.randn(slots, ctx, Hkv, D).half()
    q = torch.randn(B, 1, Hq, D).half()
    kn = torch.randn(B, 1, Hkv, D).half()
    vn = torch.randn(B, 1, Hkv, D).half()
    # the reference's decode call: ragged lengths near the context limit, slots picked by cache_batch_idx, [:, :max_len] view
    lens = torch.tensor([ctx - 1 - 37 * i for i in range(B)], dtype=torch.int32)
    idx = torch.randperm(slots)[:B].to(torch.int32)                                      # distinct slots, shuffled
    assert len(set(idx.tolist())) == B
    max_len = int(lens.max()) + 1
    kg, vg = kc.to(DEV), vc.to(DEV)
    out = flash_attn_with_kvcache(q.to(DEV), kg[:, :max_len], vg[:, :max_len], kn.to(DEV), vn.to(DEV), cache_seqlens=lens.to(DEV),
                                  cache_batch_idx=idx.to(DEV), causal=True)
    torch.cuda.synchronize()
    k1, v1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, k1[:, :max_len], v1[:, :max_len], kn, vn, cache_seqlens=lens, cache_batch_idx=idx, causal=True)
    k2, v2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, k2[:, :max_len], v2[:, :max_len], kn, vn, cache_seqlens=lens, cache_batch_idx=idx, causal=True, math="f32")
    _check(out, ref64, ref32, "decode B=%d ctx=%d" % (B, ctx))
    assert torch.equal(kg.cpu(), k1) and torch.equal(vg.cpu(), v1)        # fused append bit-exact, nothing else touched


def test_c2_decode_b16_at_32k_full_oracle():
    _decode_case(16, 32768, 32, 4, seed=2)


def test_c4_decode_b8_at_128k_full_oracle():
    _decode_case(8, 131072, 28, 4, seed=4)


def _prefill_blocks(n, c, Hq, Hkv, blocks, seed, what):
    """Whole launch on the GPU (n new tokens after c cached ones), sampled query blocks against the oracle."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    Lk = c + n
    q = torch.randn(1, n, Hq, D, device=DEV, generator=g).half()
    k = torch.randn(1, Lk + 64, Hkv, D, device=DEV, generator=g).half()      # rows past Lk exist but are not visible
    v = torch.randn(1, Lk + 64, Hkv, D, device=DEV, generator=g).half()
    cl = torch.tensor([Lk], dtype=torch.int32, device=DEV)
    out = flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, causal=True, _max_seqlen_k=Lk)
    torch.cuda.synchronize()
    kc, vc = k.cpu(), v.cpu()
    for a, b in blocks:
        qs = q[:, a:b].cpu()
        ref64 = flash_attn_with_kvcache_ref(qs, kc, vc, cache_seqlens=c + b, causal=True)
        ref32 = flash_attn_with_kvcache_ref(qs, kc, vc, cache_seqlens=c + b, causal=True, math="f32")
        _check(out[:, a:b], ref64, ref32, "%s rows [%d, %d)" % (what, a, b))


def test_c2_whole_prompt_32702_sampled_blocks_vs_oracle():
    n = 32702
    _prefill_blocks(n, 0, 32, 4, [(0, 256), (16384, 16640), (n - 256, n)], seed=22, what="c2 prefill n=32702")


def test_c4_16k_chunk_at_112k_sampled_blocks_vs_oracle():
    n, c = 16384, 130810 - 16384          # the last Sarathi chunk of the 130 810-token prompt (run_figure_6.sh:32-33)
    _prefill_blocks(n, c, 28, 4, [(0, 128), (8192, 8320), (n - 128, n)], seed=44, what="c4 16k chunk @ 112k")
