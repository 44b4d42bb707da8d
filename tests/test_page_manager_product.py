"""The product's C++ page manager (libvattn_amd.so through the C ABI, fake physical backend)
against the reference's recorded answers and the Python oracle.  Bit-exact bookkeeping."""
import os

import pytest

from oracle import trace as T
from oracle.pagemgr import PageManagerOracle
from tests.golden_util import load, pagemgr_files
from tests.impls import ProductImpl, fake_counters

FILES = pagemgr_files()
KEYS = ("ret", "err", "mapped", "lens", "pool", "pool_handles", "pagemap")


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[8:-8] for p in FILES])
@pytest.mark.parametrize("flags", [4, 0], ids=["inline", "mapper_thread"])
def test_product_matches_reference_golden(path, flags):
    g = load(path)
    cfg = g["config"]
    for tr in g["traces"]:
        impl = ProductImpl(cfg, flags=flags)
        got = T.replay(impl, tr["ops"], full=True)
        for i, (a, b) in enumerate(zip(got, tr["expect"])):
            for k in KEYS:
                assert a[k] == b[k], "trace %s/%s op %d %s key %s" % (tr["kind"], tr["seed"], i, tr["ops"][i][:1], k)
        c = fake_counters()
        assert c["violations"] == 0
        if tr["ops"][-1][0] == "cleanup":
            assert c["mapped_pages"] == 0 and c["live_handles"] == 0 and c["reserved_ranges"] == 0
        impl.pm.close()


def test_physical_mappings_track_bookkeeping():
    """After every op the set of (tensor, offset) pages mapped in the backend equals the oracle's."""
    cfg = dict(num_layers=3, num_kv_heads=4, head_size=128, max_batch_size=6, max_context_length=4096,
               itemsize=2, page_size=64 << 10, megacache=False)
    for seed in range(3):
        tr = T.gen_serving_trace(cfg, 300 + seed, iters=60, pool_groups=[40, 12, 25][seed], use_async=seed != 1,
                                 chunk=[0, 512, 128][seed], p_finish=0.05, disable_deferred=seed == 2)
        ops = T.resolve(tr, T.OracleImpl)
        o = T.OracleImpl(cfg)
        p = ProductImpl(cfg, flags=0)
        for op in ops[:-1]:
            ra = T.replay(o, [op])
            rb = T.replay(p, [op])
            assert ra == rb
            assert p.mapped_ranges() == o.o.mapped_ranges()
            assert fake_counters()["stale_vas"] == 0      # every unmap was followed by a TLB invalidation before the call returned / the batch completed
        st = p.pm.stats()
        assert st["access_calls"] <= st["map_calls"]
        assert fake_counters()["violations"] == 0
        p.pm.cleanup()
        p.pm.close()


def test_access_merge_reduces_calls():
    cfg = dict(num_layers=2, num_kv_heads=8, head_size=128, max_batch_size=4, max_context_length=32768,
               itemsize=2, page_size=64 << 10, megacache=False)
    counts = {}
    for flags in (4, 4 | 2):
        p = ProductImpl(cfg, flags=flags)
        p.reserve_physical_pages(4000 * (64 << 10))
        s = p.alloc_new_batch_idx(30000)
        lens = [0] * 4
        lens[s] = 30000
        p.step_async(lens)
        st = p.pm.stats()
        counts[flags] = (st["map_calls"], st["access_calls"])
        p.pm.cleanup()
        p.pm.close()
    assert counts[4][0] == counts[6][0]
    assert counts[6][1] == counts[6][0]            # unmerged: one set-access per page
    assert counts[4][1] <= 2 * 2 * 2               # merged: one per tensor per (sync, async) batch


def test_invalid_configs_are_explicit_errors():
    """vattention.cu:107-110 asserts (compiled out under NDEBUG in the reference) and the granularity
    assert (cudaInternal.h:33) become ValueError; seq_lens length is validated."""
    good = dict(num_layers=2, num_kv_heads=8, head_size=128, max_batch_size=4, max_context_length=4096,
                itemsize=2, page_size=2 << 20, megacache=False)
    for k, v in [("max_batch_size", 1000), ("max_context_length", 1000000), ("num_layers", 100), ("num_kv_heads", 256),
                 ("page_size", 6000), ("page_size", 1024)]:
        bad = dict(good)
        bad[k] = v
        with pytest.raises(ValueError):
            ProductImpl(bad)
    p = ProductImpl(good)
    with pytest.raises(ValueError):
        p.step_async([0, 0])
    with pytest.raises(ValueError):
        p.step([0] * 5, True)
    p.pm.close()


def test_padded_request_stride_is_explicit():
    """per-request stride is the page-rounded size, visible in the layout, never silent."""
    cfg = dict(num_layers=1, num_kv_heads=7, head_size=128, max_batch_size=3, max_context_length=1000,
               itemsize=2, page_size=2 << 20, megacache=False)    # 1.75 MB of rows -> rounded to one 2 MiB page
    p = ProductImpl(cfg)
    lay = p.pm.layout
    assert lay.virt_bytes_per_req == 2 << 20
    assert p.pm.shape() == [3, 1000, 7, 128]
    assert p.pm.stride() == [(2 << 20) // 2, 7 * 128, 128, 1]
    o = PageManagerOracle(1, 7, 128, 3, 1000, 2, 2 << 20, False)
    assert o.virt_buff_size_per_req == lay.virt_bytes_per_req and o.tokens_per_page == lay.tokens_per_page
    p.pm.close()


def test_oom_in_step_async_is_an_exception_not_a_crash():
    cfg = dict(num_layers=2, num_kv_heads=8, head_size=128, max_batch_size=4, max_context_length=16384,
               itemsize=2, page_size=2 << 20, megacache=False)
    p = ProductImpl(cfg)
    p.reserve_physical_pages(2 * 2 * 2 * (2 << 20))          # two page-groups
    s = p.alloc_new_batch_idx(9000)
    lens = [0] * 4
    lens[s] = 9000                                            # needs 9 groups
    with pytest.raises(RuntimeError, match="OOM on demand"):
        p.step_async(lens)
    o = T.OracleImpl(cfg)
    o.reserve_physical_pages(2 * 2 * 2 * (2 << 20))
    o.alloc_new_batch_idx(9000)
    with pytest.raises(RuntimeError, match="OOM on demand"):
        o.step_async(lens)
    assert p.snapshot() == o.snapshot()
    p.pm.close()


def test_lazy_pool_and_driver_failure_surface():
    from tests.impls import fake
    cfg = dict(num_layers=1, num_kv_heads=8, head_size=128, max_batch_size=2, max_context_length=8192,
               itemsize=2, page_size=2 << 20, megacache=False)
    p = ProductImpl(cfg, flags=4)                              # inline: no background pre-creation
    assert p.reserve_physical_pages(64 << 20) == 32
    assert fake_counters()["n_create"] == 0                    # nothing materialised yet
    fake().vattn_fake_fail_create_after(4)
    s = p.alloc_new_batch_idx(8000)                            # 8 groups -> 16 handles, creation fails after 4
    with pytest.raises(RuntimeError):
        p.step([8000, 0] if s == 0 else [0, 8000], False)
    p.pm.close()


@pytest.mark.parametrize("flags", [0, 4], ids=["mapper_thread", "inline"])
def test_lifecycle_cleanup_twice_use_after_cleanup_destroy_with_pending_work(flags):
    """Teardown paths of the C ABI: cleanup is idempotent, mutating calls after cleanup are explicit errors (the reference
    dereferences freed state), destroying a manager whose mapper thread still has a batch queued joins it and releases
    every handle and reservation."""
    cfg = dict(num_layers=2, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=1024, itemsize=2,
               page_size=65536, megacache=False)
    p = ProductImpl(cfg, flags=flags)
    assert p.reserve_physical_pages(40 * 65536) == 40
    assert p.alloc_new_batch_idx(100) == 0
    p.step_async([300, 0, 0, 0])
    p.cleanup()
    c = fake_counters()
    assert c["violations"] == 0 and c["live_handles"] == 0 and c["mapped_pages"] == 0 and c["reserved_ranges"] == 0
    p.cleanup()                                        # second cleanup: no-op
    for call in (lambda: p.step_async([1, 0, 0, 0]), lambda: p.step([1, 0, 0, 0], True), lambda: p.reserve_physical_pages(8 * 65536)):
        with pytest.raises(ValueError):
            call()
    assert p.alloc_new_batch_idx(5) == -1
    assert fake_counters()["violations"] == 0
    q = ProductImpl(cfg, flags=flags)
    q.reserve_physical_pages(40 * 65536)
    q.step_async([500, 400, 0, 0])                     # look-ahead work is queued for the mapper thread
    q.pm.close()                                       # vattn_destroy without cleanup(): must join, unmap, release, free
    c = fake_counters()
    assert c["violations"] == 0 and c["live_handles"] == 0 and c["mapped_pages"] == 0 and c["reserved_ranges"] == 0
