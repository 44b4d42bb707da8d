"""GPU: hardware-layout assumptions of the kernels, and the HIP VMM backend of the page manager."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_library_loaded():
    from vattention_amd import _lib
    assert _lib.lib() is not None                      # no fallback: missing .so raises ImportError


def test_mfma_and_transpose_read_layouts():
    from vattention_amd import kernels
    rc, detail = kernels.selftest_layouts(torch.device("cuda:0"))
    assert rc == 0, "layout self-test failed: %s (%s)" % (detail, kernels.last_error())


def test_vmm_granularity_and_small_pages():
    from vattention_amd import vattention
    torch.zeros(1, device="cuda")
    mn, rec = vattention.granularity(0)
    print("HIP VMM granularity: min=%d recommended=%d" % (mn, rec))
    assert mn > 0 and rec % mn == 0
    assert (2 << 20) % mn == 0


def _page(mn):
    return 64 << 10 if (64 << 10) % mn == 0 else 2 << 20


def test_allocator_touch_and_bookkeeping_vs_oracle():
    """Mirror of the reference demos (microbenchmarks/vattn_samples/utils.py:43-49): after every step
    touch every live token range; bookkeeping must equal the oracle's."""
    import random
    from oracle.pagemgr import PageManagerOracle
    from vattention_amd import vattention
    torch.zeros(1, device="cuda")
    mn, _ = vattention.granularity(0)
    page = _page(mn)
    L, kvh, D, B, ctx = 2, 2, 128, 6, 4096
    ts = vattention.init_kvcache(L, kvh, D, B, ctx, 0, torch.float16, page, False)
    try:
        assert len(ts) == 2 * L and ts[0].shape == (B, ctx, kvh, D) and ts[0].device == torch.device("cuda:0")
        o = PageManagerOracle(L, kvh, D, B, ctx, 2, page, False)
        budget = 200 * 2 * L * page
        assert vattention.reserve_physical_pages(budget) == o.reserve_physical_pages(budget)
        rng = random.Random(0)
        lens = [0] * B
        for it in range(60):
            assert vattention.num_free_kvblocks() == o.num_free_kvblocks()
            if rng.random() < 0.4:
                n = rng.randrange(1, ctx // 2)
                s = vattention.alloc_new_batch_idx(n)
                assert s == o.alloc_new_batch_idx(n)
                if s >= 0:
                    lens[s] = n
            if it % 3 == 0:
                vattention.step(list(lens), True)
                o.step(list(lens), True)
            else:
                vattention.step_async(list(lens))
                o.step_async(list(lens))
            st = vattention.state()
            assert st["mapped"] == o.mapped_pages and st["lens"] == o.curr_seq_lengths and st["pool"] == len(o.pool)
            for r in range(B):                      # touch the live prefix of every layer's K and V
                if lens[r]:
                    for t in ts:
                        t[r, :lens[r]].fill_(1.0)
            torch.cuda.synchronize()
            for r in range(B):
                if lens[r]:
                    assert float(ts[0][r, lens[r] - 1, 0, 0]) == 1.0
                    lens[r] = min(ctx, lens[r] + rng.randrange(1, 40))
                    if rng.random() < 0.1:
                        vattention.free_batch_idx(r)
                        o.free_batch_idx(r)
                        lens[r] = 0
        # mapped tokens are readable up to the page boundary, and data survives later mappings
        s = vattention.stats()
        print("vmm stats", s)
        assert s["map_calls"] > 0 and s["access_calls"] <= s["map_calls"]
    finally:
        vattention.cleanup()


def test_megacache_layout_and_views():
    from vattention_amd import vattention
    torch.zeros(1, device="cuda")
    mn, _ = vattention.granularity(0)
    page = 2 << 20
    L, kvh, D, B, ctx = 4, 2, 128, 3, 2048
    ts = vattention.init_kvcache(L, kvh, D, B, ctx, 0, torch.float16, page, True)
    try:
        assert len(ts) == 2 and ts[0].shape == (B, ctx, L, kvh, D)
        vattention.reserve_physical_pages(64 * page)
        s = vattention.alloc_new_batch_idx(1500)
        lens = [0] * B
        lens[s] = 1500
        vattention.step_async(lens)
        for l in range(L):                           # per-layer views as the cache engine builds them (:58-68)
            k_l = ts[0][:, :, l]
            k_l[s, :1500].fill_(float(l + 1))
        torch.cuda.synchronize()
        assert float(ts[0][s, 1499, 2, 1, 5]) == 3.0
    finally:
        vattention.cleanup()


def test_sync_oom_raises_runtime_error():
    from vattention_amd import vattention
    torch.zeros(1, device="cuda")
    page = 2 << 20
    vattention.init_kvcache(1, 8, 128, 2, 16384, 0, torch.float16, page, False)
    try:
        vattention.reserve_physical_pages(4 * page)          # 2 groups
        s = vattention.alloc_new_batch_idx(8000)             # needs 8
        lens = [0, 0]
        lens[s] = 8000
        with pytest.raises(RuntimeError, match="OOM on demand"):
            vattention.step_async(lens)
    finally:
        vattention.set_verbose(False)
        vattention.cleanup()


def test_remap_same_va_is_visible_to_kernels():
    """ROCm 7.2 / gfx950 keeps stale GPU translations after hipMemUnmap (+ hipMemMap of another handle at
    the same VA) until the driver services an allocation (tools/remap_probe4.cpp).  The manager issues a
    TLB invalidation after every batch that unmapped something; without it this test reads stale data and
    writes through stale translations corrupt pages that moved to other slots."""
    from vattention_amd import vattention
    torch.zeros(1, device="cuda")
    for page in (64 << 10, 2 << 20):
        L, kvh, D, B, ctx = 2, 2, 128, 4, 4096
        ts = vattention.init_kvcache(L, kvh, D, B, ctx, 0, torch.float16, page, False)
        try:
            groups = 4096 * kvh * D * 2 // page          # page-groups for one full-length request
            vattention.reserve_physical_pages(2 * groups * 2 * L * page)      # room for exactly two full requests
            n = 4000
            for it in range(6):
                # slots 0 and 1 alternately own the physical pages: eager reclaim unmaps the idle slot, LIFO pool
                # order hands its pages to the other one -> same VAs, different physical pages every iteration
                a, b_ = it & 1, (it & 1) ^ 1
                lens = [0] * B
                lens[a] = n
                vattention.step(lens, True)
                va = float(10 + it)
                for t in ts:
                    t[a, :n].fill_(va)
                lens[b_] = n
                vattention.step(lens, True)
                vb = float(100 + it)
                for t in ts:
                    t[b_, :n].fill_(vb)
                torch.cuda.synchronize()
                for t in ts:                              # device-side reads (reduction kernels) of both slots
                    assert float(t[a, :n].float().min()) == va and float(t[a, :n].float().max()) == va
                    assert float(t[b_, :n].float().min()) == vb and float(t[b_, :n].float().max()) == vb
                    assert float(t[a, n - 1, 1, 7].item()) == va          # and through a D2H copy
                vattention.step([0] * B, True)            # unmap everything
            assert vattention.stats()["tlb_flushes"] > 0
        finally:
            vattention.cleanup()
