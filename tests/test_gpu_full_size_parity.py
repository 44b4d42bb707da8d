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


# ---- the shapes bench.py --gpus 8 and the `dynamic` legs time (round 3): one TP=8 rank of Llama-3-70B (8 query / 1 kv head), and
# ---- Llama-3-8B (32 / 8) over MEGACACHE views with 8 MiB pages ----

def _workspace_bytes(q, k, v, Lk, causal=True):
    """vattn_attn_workspace_bytes of the DEFAULT plan for this call (> 0 = the key range is split: KV-split kernel + combine_rows)."""
    import ctypes as C
    from vattention_amd import kernels as K
    p = K.AttnParams()
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = q.shape[0], q.shape[1], k.shape[1], q.shape[2], k.shape[2], q.shape[3]
    p.is_causal, p.dtype, p.num_splits, p.variant, p.max_seqlen_k_hint = int(causal), 0, 0, 0, Lk
    return int(K.klib().vattn_attn_workspace_bytes(C.byref(p)))


def test_tp8_rank_whole_prompt_29092_default_plan_vs_oracle():
    """The longest prompt of the dynamic trace (29 092 tokens, tests/golden/c3_arxiv_lengths_256.json) on one TP=8 rank of
    Llama-3-70B: 8 query heads on ONE kv head, default plan."""
    n = 29092
    _prefill_blocks(n, 0, 8, 1, [(0, 256), (14336, 14592), (n - 256, n)], seed=70, what="TP8 rank whole prompt n=29092")


def test_tp8_rank_2k_chunk_at_30k_takes_the_kv_split_plan_vs_oracle():
    """A 2 k chunk on a 30 k prefix, 8 / 1 heads: 64 query blocks of 256 rows cannot fill 256 CUs, so the DEFAULT plan must split the
    key range (prefill64 KV-split partials + combine_rows_kernel) — asserted through the workspace query — and the merged result
    must match the oracle."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    n, c = 2048, 30000
    g = torch.Generator(device=DEV)
    g.manual_seed(71)
    Lk = c + n
    q = torch.randn(1, n, 8, D, device=DEV, generator=g).half()
    k = torch.randn(1, Lk + 64, 1, D, device=DEV, generator=g).half()
    v = torch.randn(1, Lk + 64, 1, D, device=DEV, generator=g).half()
    assert _workspace_bytes(q, k, v, Lk) > 0, "the default plan did not split the key range of an underfilled grid"
    cl = torch.tensor([Lk], dtype=torch.int32, device=DEV)
    out = flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, causal=True, _max_seqlen_k=Lk)
    torch.cuda.synchronize()
    kc, vc = k.cpu(), v.cpu()
    for a, b in [(0, 256), (896, 1152), (n - 256, n)]:
        qs = q[:, a:b].cpu()
        ref64 = flash_attn_with_kvcache_ref(qs, kc, vc, cache_seqlens=c + b, causal=True)
        ref32 = flash_attn_with_kvcache_ref(qs, kc, vc, cache_seqlens=c + b, causal=True, math="f32")
        _check(out[:, a:b], ref64, ref32, "TP8 rank 2k chunk @ 30k rows [%d, %d)" % (a, b))


def test_tp8_rank_decode_b64_at_32k_full_oracle():
    _decode_case(64, 32768, 8, 1, seed=72)


def _megacache_engine(B, ctx, L, Hkv, page, pool_bytes):
    from vattention_amd import vattention
    vattention.enable_layered_async(False)
    ts = vattention.init_kvcache(L, Hkv, D, B, ctx, 0, torch.float16, page, True)
    vattention.reserve_physical_pages(pool_bytes)
    return vattention, ts[0], ts[1]


def test_llama8b_batch256_decode_ragged_4k_32k_over_megacache_views_vs_oracle():
    """configs[2]'s decode shape as the `dynamic` leg runs it: Llama-3-8B heads (32 / 8), 256 sequences with ragged contexts between
    4 k and 32 k, K/V = per-layer VIEWS k[:, :, l] of megacache tensors [B, ctx, L, kvh, D] backed by the real page manager with
    8 MiB pages (mapped only under each slot's prefix), fused append, slots through cache_batch_idx.  Every sequence against the
    oracle (one oracle call per sequence: the caches are virtual, only the prefixes exist)."""
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.zeros(1, device=DEV)
    B, ctx, L, Hq, Hkv, page, layer = 256, 32768, 2, 32, 8, 8 << 20, 1
    rng = torch.Generator().manual_seed(256)
    lens = torch.randint(4096, ctx - 1, (B,), generator=rng, dtype=torch.int32)
    lens[0], lens[1] = ctx - 1, 4096
    va, kmega, vmega = _megacache_engine(B, ctx, L, Hkv, page, int(lens.sum().item() + B * 2048) * L * Hkv * D * 2 * 2 + (64 * page))
    try:
        slot_of = torch.randperm(B).to(torch.int32)                    # sequence i lives in slot slot_of[i]
        step_lens = [0] * B
        for i in range(B):
            step_lens[int(slot_of[i])] = int(lens[i]) + 1
        va.step(step_lens, False)
        k_l, v_l = kmega[:, :, layer], vmega[:, :, layer]
        g = torch.Generator(device=DEV)
        g.manual_seed(257)
        host = []
        for i in range(B):                                             # fill each prefix through the product's own append
            n = int(lens[i])
            kk = torch.randn(n, Hkv, D, device=DEV, generator=g).half()
            vv = torch.randn(n, Hkv, D, device=DEV, generator=g).half()
            s = int(slot_of[i])
            cache_flat(kk, vv, k_l[s], v_l[s], "auto")
            host.append((kk.cpu(), vv.cpu()))
        q = torch.randn(B, 1, Hq, D, device=DEV, generator=g).half()
        kn = torch.randn(B, 1, Hkv, D, device=DEV, generator=g).half()
        vn = torch.randn(B, 1, Hkv, D, device=DEV, generator=g).half()
        ml = int(lens.max()) + 1
        out = flash_attn_with_kvcache(q, k_l[:, :ml], v_l[:, :ml], kn, vn, cache_seqlens=lens.to(DEV), cache_batch_idx=slot_of.to(DEV), causal=True)
        torch.cuda.synchronize()
        qh, knh, vnh, outh = q.cpu(), kn.cpu(), vn.cpu(), out.cpu()
        worst = 0.0
        # one launch computes all 256 sequences; the oracle (0.27 s of float64 per sequence) checks every FOURTH one plus the longest
        # and the shortest (VATTN_FULL_ORACLE=1: all of them, as rounds 3-5 did: 70 s)
        import os
        every = 1 if os.environ.get("VATTN_FULL_ORACLE") == "1" else 4
        for i in sorted(set(range(0, B, every)) | {0, 1}):
            kf = torch.cat([host[i][0], knh[i]]).unsqueeze(0)
            vf = torch.cat([host[i][1], vnh[i]]).unsqueeze(0)
            n1 = kf.shape[1]
            ref64 = flash_attn_with_kvcache_ref(qh[i:i + 1], kf, vf, cache_seqlens=n1, causal=True)
            ref32 = flash_attn_with_kvcache_ref(qh[i:i + 1], kf, vf, cache_seqlens=n1, causal=True, math="f32")
            _check(outh[i:i + 1], ref64, ref32, "megacache decode sequence %d of 256 (context %d)" % (i, n1))
            worst = max(worst, (outh[i:i + 1].double() - ref64).abs().max().item())
            s = int(slot_of[i])
            if i % 32 == 0:                                            # the appended row landed in this layer's rows, bit-exact
                assert torch.equal(k_l[s, n1 - 1].cpu(), knh[i, 0]) and torch.equal(v_l[s, n1 - 1].cpu(), vnh[i, 0])
        print("batch-256 megacache decode: worst abs err %.3e" % worst)
    finally:
        va.cleanup()


def test_llama8b_varlen_prefill_4_prompts_over_megacache_views_vs_oracle():
    """One vLLM-scheduler iteration of the `dynamic` leg: four whole prompts (4-12 k tokens, 28 k tokens together) of Llama-3-8B
    (32 / 8 heads) in ONE batched variable-length launch over megacache views with 8 MiB pages; sampled query blocks of every
    prompt against the oracle."""
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache
    torch.zeros(1, device=DEV)
    B, ctx, L, Hq, Hkv, page, layer = 8, 32768, 2, 32, 8, 8 << 20, 0
    prompts = [4119, 7344, 12001, 5000]
    slots = [5, 0, 3, 6]
    va, kmega, vmega = _megacache_engine(B, ctx, L, Hkv, page, (sum(prompts) + 4 * 4096) * L * Hkv * D * 2 * 2 + 64 * page)
    try:
        step_lens = [0] * B
        for s, n in zip(slots, prompts):
            step_lens[s] = n
        va.step(step_lens, False)
        k_l, v_l = kmega[:, :, layer], vmega[:, :, layer]
        g = torch.Generator(device=DEV)
        g.manual_seed(88)
        T = sum(prompts)
        q = torch.randn(T, Hq, D, device=DEV, generator=g).half()
        host, tok = [], 0
        for s, n in zip(slots, prompts):
            kk = torch.randn(n, Hkv, D, device=DEV, generator=g).half()
            vv = torch.randn(n, Hkv, D, device=DEV, generator=g).half()
            cache_flat(kk, vv, k_l[s], v_l[s], "auto")
            host.append((kk.cpu(), vv.cpu()))
        starts = torch.tensor([sum(prompts[:i]) for i in range(4)], dtype=torch.int32, device=DEV)
        qlens = torch.tensor(prompts, dtype=torch.int32, device=DEV)
        out = torch.full((T, Hq, D), float("nan"), dtype=torch.float16, device=DEV)
        flash_attn_varlen_with_kvcache(q, k_l, v_l, starts, qlens, max(prompts), qlens, torch.tensor(slots, dtype=torch.int32, device=DEV),
                                       causal=True, out=out, _max_seqlen_k=max(prompts))
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any(), "rows left unwritten by the batched launch"
        qh, oh = q.cpu(), out.cpu()
        for i, n in enumerate(prompts):
            kf, vf = host[i][0].unsqueeze(0), host[i][1].unsqueeze(0)
            for a, b in [(0, 128), (n // 2, n // 2 + 128), (n - 128, n)]:
                qs = qh[tok + a:tok + b].unsqueeze(0)
                ref64 = flash_attn_with_kvcache_ref(qs, kf, vf, cache_seqlens=b, causal=True)
                ref32 = flash_attn_with_kvcache_ref(qs, kf, vf, cache_seqlens=b, causal=True, math="f32")
                _check(oh[tok + a:tok + b].unsqueeze(0), ref64, ref32, "varlen prompt %d (%d tokens) rows [%d, %d)" % (i, n, a, b))
            tok += n
    finally:
        va.cleanup()


def test_prefill_chunk_over_a_layer_view_that_spans_more_than_4_gib_vs_oracle():
    """A layer's view k[:, :, l] of a megacache tensor [slots, tokens, layers, heads, d] has rows of layers x heads x 256 bytes: at
    64 KiB per row 70 000 tokens span 4.3 GiB, more than a 32-bit buffer bound holds.  The prefill kernel's running descriptors count
    ROWS (prefill64_kernels.hip, k_rsrc_advance); this drives a 256-row chunk on a 69 744-token prefix through them, and the last
    rows of the sequence lie beyond the 4 GiB mark."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(41)
    L, Hkv, Hq, n, c = 32, 8, 8, 256, 69744
    Lk = n + c
    kmega = torch.empty(1, Lk, L, Hkv, D, dtype=torch.float16, device=DEV)
    vmega = torch.empty(1, Lk, L, Hkv, D, dtype=torch.float16, device=DEV)
    assert kmega.stride(1) * 2 == 64 << 10 and Lk * kmega.stride(1) * 2 > (1 << 32)
    layer = 17
    kmega[:, :, layer].normal_()
    vmega[:, :, layer].normal_()
    kv, vv = kmega[:, :, layer], vmega[:, :, layer]                      # [1, Lk, Hkv, D], row stride 64 KiB
    q = torch.randn(1, n, Hq, D).half()
    cl = torch.tensor([Lk], dtype=torch.int32)
    out = flash_attn_with_kvcache(q.to(DEV), kv, vv, cache_seqlens=cl.to(DEV), causal=True)
    torch.cuda.synchronize()
    kc, vc = kv.cpu().contiguous(), vv.cpu().contiguous()
    ref64 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, causal=True, math="f32")
    _check(out, ref64, ref32, "256-row chunk on a 69 744-token prefix over a 64 KiB-row view")
