"""Seeded random sweep of the operator's argument space against the CPU oracle (small sizes, many shapes): head counts and
group sizes, head dimension 64 / 128, fp16 / bf16, ragged batches through cache_batch_idx, causal and full attention,
query chunks longer than the visible keys, forced and automatic split counts, every workgroup order / tiling, batched
variable-length chunks, decode with in-kernel append.  Deterministic (fixed seeds), ~100 cases."""
import os
import random

import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref
from tests.variants import product_or_self

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# VATTN_FUZZ_SCALE=10 runs ten times as many seeds (the committed default keeps the suite short)
SCALE = int(os.environ.get("VATTN_FUZZ_SCALE", "1"))
BASE = int(os.environ.get("VATTN_FUZZ_SEED_BASE", "0"))      # shifts every seed: a different sample of the same space


def _tol(dtype):
    return (2e-3, 2e-3) if dtype == torch.float16 else (1.6e-2, 1.6e-2)


def _check(out, ref64, dtype, what):
    atol, rtol = _tol(dtype)
    err = (out.double().cpu() - ref64).abs()
    bound = atol + rtol * ref64.abs()
    assert bool((err <= bound).all()), "%s: max err %.3e" % (what, err.max().item())


@pytest.mark.parametrize("seed", range(BASE, BASE + 6 * SCALE))
def test_fuzz_prefill(seed):
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    rng = random.Random(1000 + seed)
    torch.manual_seed(seed)
    for case in range(8):
        D = rng.choice([64, 128, 128])
        dtype = rng.choice([torch.float16, torch.float16, torch.bfloat16])
        Hkv = rng.choice([1, 2, 3, 4, 8])
        G = rng.choice([1, 2, 4, 7, 8])
        Hq = Hkv * G
        B = rng.choice([1, 1, 2, 3])
        n = rng.choice([2, 17, 64, 100, 129, 257, 300, 520])
        ctx = 1200
        causal = rng.random() < 0.8
        cls = [rng.choice([0, 1, 30, 64, 333, 600]) + (n if rng.random() < 0.85 else rng.randrange(1, n + 1)) for _ in range(B)]
        slots = rng.sample(range(B + 2), B)
        variant = rng.choice([0, 0, 1, 2, 8, 9, 32, 64, 12 if D == 128 else 0, 4 if D == 128 else 0, 14 if D == 128 else 0, 782 if D == 128 else 2, 14 if D == 128 else 0, 526 if D == 128 else 8, 6 if D == 128 else 0, 6 if D == 128 else 2])
        splits = rng.choice([0, 0, 0, 1, 2, 3, 7])
        if splits > 1:
            variant |= rng.choice([0, 0, 16384, 32768])      # the key-range shares merged inside the launch (both protocols)
        q = torch.randn(B, n, Hq, D).to(dtype)
        kc = torch.randn(B + 2, ctx, Hkv, D).to(dtype)
        vc = torch.randn(B + 2, ctx, Hkv, D).to(dtype)
        cl = torch.tensor(cls, dtype=torch.int32)
        idx = torch.tensor(slots, dtype=torch.int32)
        ref = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=causal)
        out = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV),
                                      causal=causal, num_splits=splits, _variant=product_or_self(variant), _max_seqlen_k=rng.choice([0, max(cls)]))
        torch.cuda.synchronize()
        _check(out, ref, dtype, "prefill seed %d case %d (D=%d Hq=%d Hkv=%d B=%d n=%d cl=%s causal=%s variant=%d splits=%d)" % (
            seed, case, D, Hq, Hkv, B, n, cls, causal, variant, splits))


@pytest.mark.parametrize("seed", range(BASE, BASE + 4 * SCALE))
def test_fuzz_batched_chunks(seed):
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache
    rng = random.Random(2000 + seed)
    torch.manual_seed(100 + seed)
    for case in range(5):
        D = rng.choice([64, 128])
        Hkv = rng.choice([1, 2, 4])
        Hq = Hkv * rng.choice([1, 4, 8])
        B = rng.choice([2, 3, 5])
        lens = [rng.choice([1, 5, 64, 130, 256, 300, 511]) for _ in range(B)]
        if max(lens) < 2:
            lens[0] = 40
        pre = [rng.choice([0, 3, 128, 400]) for _ in range(B)]
        ctx = 1000
        T = sum(lens)
        q = torch.randn(T, Hq, D).half()
        kc = torch.randn(B + 1, ctx, Hkv, D).half()
        vc = torch.randn(B + 1, ctx, Hkv, D).half()
        starts = [sum(lens[:i]) for i in range(B)]
        cls = [a + b for a, b in zip(pre, lens)]
        slots = rng.sample(range(B + 1), B)
        i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
        out = flash_attn_varlen_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), i32(starts), i32(lens), max(lens), i32(cls), i32(slots),
                                             causal=True, num_splits=rng.choice([0, 0, 2, 5]), _variant=product_or_self(rng.choice([0, 2, 8, 64, 14, 782, 6])),
                                             _max_seqlen_k=max(cls))
        torch.cuda.synchronize()
        for i in range(B):
            ref = flash_attn_with_kvcache_ref(q[starts[i]:starts[i] + lens[i]].unsqueeze(0), kc, vc,
                                              cache_seqlens=torch.tensor([cls[i]], dtype=torch.int32),
                                              cache_batch_idx=torch.tensor([slots[i]], dtype=torch.int32), causal=True)
            _check(out[starts[i]:starts[i] + lens[i]].unsqueeze(0), ref, torch.float16, "varlen seed %d case %d entry %d" % (seed, case, i))


@pytest.mark.parametrize("seed", range(BASE, BASE + 5 * SCALE))
def test_fuzz_decode(seed):
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    rng = random.Random(3000 + seed)
    torch.manual_seed(200 + seed)
    for case in range(8):
        D = rng.choice([64, 128, 128])
        dtype = rng.choice([torch.float16, torch.bfloat16])
        Hkv = rng.choice([1, 2, 4, 8])
        G = rng.choice([1, 4, 7, 8, 16, 17, 40])
        Hq = Hkv * G
        B = rng.choice([1, 2, 5, 9])
        ctx = rng.choice([40, 700, 2100])
        cls = [rng.randrange(0, ctx - 1) for _ in range(B)]
        slots = rng.sample(range(B + 3), B)
        append = rng.random() < 0.7
        q = torch.randn(B, 1, Hq, D).to(dtype)
        kc = torch.randn(B + 3, ctx, Hkv, D).to(dtype)
        vc = torch.randn(B + 3, ctx, Hkv, D).to(dtype)
        kn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
        vn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
        cl = torch.tensor(cls if append else [c + 1 for c in cls], dtype=torch.int32)
        idx = torch.tensor(slots, dtype=torch.int32)
        kr, vr = kc.clone(), vc.clone()
        ref = flash_attn_with_kvcache_ref(q, kr, vr, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
        kg, vg = kc.to(DEV), vc.to(DEV)
        out = flash_attn_with_kvcache(q.to(DEV), kg, vg, kn.to(DEV) if append else None, vn.to(DEV) if append else None,
                                      cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV), causal=True,
                                      num_splits=rng.choice([0, 0, 1, 2, 9, 48]), _variant=product_or_self(rng.choice([0, 1, 64, 128, 512, 1024, 128 | 1024])))
        torch.cuda.synchronize()
        _check(out, ref, dtype, "decode seed %d case %d (D=%d Hq=%d Hkv=%d B=%d ctx=%d append=%s)" % (seed, case, D, Hq, Hkv, B, ctx, append))
        assert torch.equal(kg.cpu(), kr) and torch.equal(vg.cpu(), vr)


@pytest.mark.parametrize("seed", range(BASE, BASE + 4 * SCALE))
def test_fuzz_prefill64_midsize(seed):
    """The round-2 prefill kernel (variant 14 = forced, product build; 782 = XOR-swizzled K image) on
    shapes big enough for its pipeline to reach steady state: hundreds to thousands of query rows on prefixes of up to 7 k keys,
    ragged two-sequence batches, key-range shares (two-launch and in-launch merges), non-causal, odd group sizes."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    rng = random.Random(7000 + seed)
    torch.manual_seed(700 + seed)
    for case in range(3):
        dtype = rng.choice([torch.float16, torch.float16, torch.bfloat16])
        Hkv = rng.choice([1, 2, 4])
        G = rng.choice([1, 4, 7, 8])
        Hq, D = Hkv * G, 128
        B = rng.choice([1, 1, 2])
        n = rng.choice([257, 300, 777, 1024, 1500, 2048, 3000])
        pre = [rng.choice([0, 0, 64, 1000, 3333, 7000]) for _ in range(B)]
        cls = [p + (n if rng.random() < 0.8 else rng.randrange(1, n + 1)) for p in pre]
        ctx = max(cls) + 7
        causal = rng.random() < 0.85
        variant = rng.choice([14, 14, 14, 782, 782, 6, 6, 6])
        splits = rng.choice([0, 0, 1, 2, 3])
        if splits > 1:
            variant |= rng.choice([0, 16384, 32768])
        slots = rng.sample(range(B + 1), B)
        q = torch.randn(B, n, Hq, D).to(dtype)
        kc = torch.randn(B + 1, ctx, Hkv, D).to(dtype)
        vc = torch.randn(B + 1, ctx, Hkv, D).to(dtype)
        cl = torch.tensor(cls, dtype=torch.int32)
        idx = torch.tensor(slots, dtype=torch.int32)
        ref = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=causal)
        out = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV),
                                      causal=causal, num_splits=splits, _variant=product_or_self(variant), _max_seqlen_k=max(cls))
        torch.cuda.synchronize()
        _check(out, ref, dtype, "prefill64 seed %d case %d (Hq=%d Hkv=%d B=%d n=%d cl=%s causal=%s variant=%d splits=%d %s)" % (
            seed, case, Hq, Hkv, B, n, cls, causal, variant, splits, dtype))


@pytest.mark.parametrize("seed", range(BASE, BASE + 4 * SCALE))
def test_fuzz_planned_launches(seed):
    """Round-3 launch plans under random shapes: ragged decode batches through the length-balanced split (random piece lengths,
    d = 64 / 128, groups up to 32 heads, with and without the append) and batched prefill chunks through the work list (random piece
    lengths, prefixes, one-token entries, causal and full) — each against the oracle."""
    from vattention_amd import flash_attn as FA
    from vattention_amd import kernels as K
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache
    rng = random.Random(7000 + seed)
    torch.manual_seed(700 + seed)
    for case in range(5):
        # ---- decode ----
        D = rng.choice([64, 128, 128])
        dtype = rng.choice([torch.float16, torch.bfloat16])
        Hkv = rng.choice([1, 2, 4])
        G = rng.choice([1, 4, 7, 8, 16, 32])
        Hq = Hkv * G
        B = rng.choice([2, 3, 6, 11])
        ctx = rng.choice([300, 1500, 2600])
        cls = [rng.randrange(0, ctx - 1) for _ in range(B)]
        slots = rng.sample(range(B + 2), B)
        append = rng.random() < 0.75
        q = torch.randn(B, 1, Hq, D).to(dtype)
        kc = torch.randn(B + 2, ctx, Hkv, D).to(dtype)
        vc = torch.randn(B + 2, ctx, Hkv, D).to(dtype)
        kn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
        vn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
        lens = cls if append else [c + 1 for c in cls]
        cl = torch.tensor(lens, dtype=torch.int32)
        idx = torch.tensor(slots, dtype=torch.int32)
        kr, vr = kc.clone(), vc.clone()
        ref = flash_attn_with_kvcache_ref(q, kr, vr, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
        kg, vg = kc.to(DEV), vc.to(DEV)
        cap = []
        tiles = rng.choice([1, 2, 5, 9, 30])
        out = flash_attn_with_kvcache(q.to(DEV), kg, vg, kn.to(DEV) if append else None, vn.to(DEV) if append else None,
                                      cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV), causal=True, _cache_seqlens_host=lens,
                                      _plan_tiles=tiles, _params_out=cap)
        torch.cuda.synchronize()
        what = "planned decode seed %d case %d (D=%d Hq=%d Hkv=%d B=%d ctx=%d append=%s tiles=%d items=%d)" % (
            seed, case, D, Hq, Hkv, B, ctx, append, tiles, cap[0].num_split_items)
        if G <= 32:
            assert cap[0].num_split_items >= B, what
        _check(out, ref, dtype, what)
        assert torch.equal(kg.cpu(), kr) and torch.equal(vg.cpu(), vr), what
        # the product path (round 4): the decomposition derived ON THE DEVICE from cache_seqlens — default workgroup count and a random one
        for nwg in (0, -rng.choice([1, 2, 3, 7, 19, 64, 200])):
            kg2, vg2 = kc.to(DEV), vc.to(DEV)
            out2 = flash_attn_with_kvcache(q.to(DEV), kg2, vg2, kn.to(DEV) if append else None, vn.to(DEV) if append else None,
                                           cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV), causal=True, num_splits=nwg)
            torch.cuda.synchronize()
            _check(out2, ref, dtype, what + " device-planned stream, num_splits=%d" % nwg)
            assert torch.equal(kg2.cpu(), kr) and torch.equal(vg2.cpu(), vr), what
        # ---- prefill work list (d = 128) ----
        Hkv = rng.choice([1, 2, 4])
        Hq = Hkv * rng.choice([1, 2, 4, 8])
        P = rng.choice([1, 2, 3, 4])
        causal = rng.random() < 0.8
        chunks = [(rng.choice([0, 0, 40, 700, 1500]), rng.choice([1, 2, 60, 256, 257, 600, 1100])) for _ in range(P)]
        if all(n == 1 for _, n in chunks):
            chunks[0] = (chunks[0][0], 300)
        ctx = max(c + n for c, n in chunks) + 3
        kc = torch.randn(P + 1, ctx, Hkv, 128).to(dtype)
        vc = torch.randn(P + 1, ctx, Hkv, 128).to(dtype)
        T = sum(n for _, n in chunks)
        q = torch.randn(T, Hq, 128).to(dtype)
        sl = torch.tensor(rng.sample(range(P + 1), P), dtype=torch.int32)
        q_lens, k_lens = [n for _, n in chunks], [c + n for c, n in chunks]
        p = K.AttnParams()
        p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = P, max(q_lens), Hq, Hkv, 128, int(causal)
        tiles = rng.choice([1, 3, 8, 40])
        plan = FA.prefill_plan(p, q_lens, k_lens, torch.device(DEV), force_tiles=tiles)
        out = torch.full((T, Hq, 128), float("nan"), dtype=dtype, device=DEV)
        starts = torch.tensor([sum(q_lens[:i]) for i in range(P)], dtype=torch.int32, device=DEV)
        flash_attn_varlen_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), starts, torch.tensor(q_lens, dtype=torch.int32, device=DEV), max(q_lens),
                                       torch.tensor(k_lens, dtype=torch.int32, device=DEV), sl.to(DEV), causal=causal, out=out,
                                       _max_seqlen_k=max(k_lens), _pf_plan=plan)
        torch.cuda.synchronize()
        what = "work-list prefill seed %d case %d (Hq=%d Hkv=%d chunks=%s causal=%s tiles=%d pieces=%d split blocks=%d %s)" % (
            seed, case, Hq, Hkv, chunks, causal, tiles, plan.n_items, plan.n_blocks, dtype)
        assert plan.t is not None and not torch.isnan(out.float()).any(), what
        tok = 0
        for i, (c, n) in enumerate(chunks):
            s_ = int(sl[i])
            ref = flash_attn_with_kvcache_ref(q[tok:tok + n].unsqueeze(0), kc[s_:s_ + 1].clone(), vc[s_:s_ + 1].clone(),
                                              cache_seqlens=torch.tensor([c + n], dtype=torch.int32), causal=causal)
            _check(out[tok:tok + n].unsqueeze(0), ref, dtype, what + " entry %d" % i)
            tok += n
