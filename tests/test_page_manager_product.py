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
        seen_common = False
        for i, (a, b) in enumerate(zip(got, tr["expect"])):
            seen_common |= tr["ops"][i][0] == "map_common"
            for k in KEYS:
                if k == "ret" and seen_common and tr["ops"][i][0] == "nfree":
                    # deliberate fix: a page-group aliased into B slots is ONE group of physical pages; the reference counts it
                    # B times (utils.h:177-183 sums mapped - needed per slot); see test_shared_prefix_pages_are_refcounted
                    assert a[k] <= b[k]
                    continue
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


def test_wait_pool_ready_returns_once_the_mapper_has_created_the_pool():
    """vattn_wait_pool_ready (round 5): a small pool (<= 40 000 pages) is created WHOLE by the idle mapper thread; the call returns 0
    once every handle exists (what the reference's reserve_physical_pages guarantees, cudaInternal.h:45-59) — and at once, with 0,
    where nothing is created ahead of demand (inline mode)."""
    cfg = dict(num_layers=2, num_kv_heads=8, head_size=128, max_batch_size=4, max_context_length=8192,
               itemsize=2, page_size=2 << 20, megacache=False)
    p = ProductImpl(cfg, flags=0)                              # mapper thread
    assert p.reserve_physical_pages(256 << 20) == 128
    assert p.pm.wait_pool_ready(-1) == 0
    assert fake_counters()["n_create"] == 128                  # the whole pool, none on demand later
    s = p.alloc_new_batch_idx(4000)
    p.step([4000 if i == s else 0 for i in range(4)], False)
    assert fake_counters()["n_create"] == 128
    p.pm.close()
    q = ProductImpl(cfg, flags=4)                              # inline: lazy, no mapper
    assert q.reserve_physical_pages(256 << 20) == 128
    assert q.pm.wait_pool_ready(0) == 0 and fake_counters()["n_create"] == 0
    q.pm.close()


def test_wait_pool_ready_reports_a_failed_creation_ahead_of_demand():
    """a hipMemCreate that fails while the idle mapper creates the pool must not read as "ready" — the reference aborts
    inside reserve (cudaInternal.h:45-59, CHECK_CUDA); here wait_pool_ready returns VATTN_ERR_DRIVER and the reserve drop-in raises."""
    from tests.impls import fake
    from vattention_amd import _lib as L
    cfg = dict(num_layers=2, num_kv_heads=8, head_size=128, max_batch_size=4, max_context_length=8192,
               itemsize=2, page_size=2 << 20, megacache=False)
    p = ProductImpl(cfg, flags=0)
    fake().vattn_fake_fail_create_after(10)                    # the device "runs out" after 10 handles
    assert p.reserve_physical_pages(256 << 20) == 128
    assert p.pm.wait_pool_ready(5000) == L.VATTN_ERR_DRIVER
    assert fake_counters()["n_create"] <= 11
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


# ---------------------------------------------------------------------------------------------------------------------
# round 2: prefix-sharing refcounts, rollback after a driver failure, per-slot fences, layer-ordered mapping
# ---------------------------------------------------------------------------------------------------------------------
SHARE_CFG = dict(num_layers=2, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=2048, itemsize=2,
                 page_size=32 << 10, megacache=False)          # 64 tokens per page


def test_shared_prefix_pages_are_refcounted():
    """map_common_pages aliases ONE physical pair per layer into every slot (vattention.cu:325-373).  The reference returns
    such a page to the pool once per slot when slots are reclaimed (mux.h:51-66): duplicate ids in the LIFO pool, i.e. the same
    physical page handed to two different (slot, layer) ranges later.  The product keeps a per-page refcount: the page goes back
    exactly once, when its last mapping is unmapped, and num_free_kvblocks counts the shared group once."""
    from tests.impls import fake_mapped
    cfg = SHARE_CFG
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    # the reference's behaviour, restated by the oracle: duplicates
    ref = T.OracleImpl(cfg)
    ref.reserve_physical_pages(10 * group)
    ref.map_common_pages(100)                       # 2 page-groups shared by all 4 slots
    ref.step([0, 0, 0, 0], True)                    # eager reclaim of every slot
    assert len(ref.o.pool) > len(set(ref.o.pool))   # the reference bug this test documents

    p = ProductImpl(cfg, flags=4)
    o = T.OracleImpl(cfg, shared_page_refcount=True)
    for impl in (p, o):
        impl.reserve_physical_pages(10 * group)
        impl.map_common_pages(100)
    st = p.pm.state()
    assert st["mapped"] == [2, 2, 2, 2] and st["pool"] == 10 * 4 - 2 * 4
    assert p.num_free_kvblocks() == o.num_free_kvblocks() == 8 + 2      # 8 pool groups + the 2 shared ones, ONCE each
    assert len({h for _va, _n, h, _a in fake_mapped()}) == 8               # 8 physical pages behind 32 mappings
    # a request that covers the prefix pins it: nothing of the shared groups is free-able
    s = p.alloc_new_batch_idx(150)
    assert s == o.alloc_new_batch_idx(150) == 0
    lens = [150, 0, 0, 0]
    p.step(lens, False); o.step(lens, False)
    assert p.num_free_kvblocks() == o.num_free_kvblocks() == 7           # one exclusive group mapped for tokens 128..149
    # eager reclaim of the idle slots drops 3 of the 4 mappings of every shared page: nothing returns to the pool yet
    p.step(lens, True); o.step(lens, True)
    st = p.pm.state()
    assert st["mapped"] == [3, 0, 0, 0] and st["pool"] == o.snapshot()["pool"] == 7 * 4
    assert len(st["pool_ids"]) == len(set(st["pool_ids"]))
    # last holder goes: the shared pages return, exactly once
    p.free_batch_idx(0); o.free_batch_idx(0)
    p.step([0, 0, 0, 0], True); o.step([0, 0, 0, 0], True)
    st = p.pm.state()
    assert st["mapped"] == [0, 0, 0, 0] and st["pool"] == 10 * 4
    assert sorted(st["pool_ids"]) == list(range(40))                      # every id once
    assert p.snapshot(True)["pool_handles"] == o.snapshot(True)["pool_handles"]
    c = fake_counters()
    assert c["violations"] == 0 and c["mapped_pages"] == 0 and c["stale_vas"] == 0
    p.pm.cleanup(); p.pm.close()


@pytest.mark.parametrize("flags", [4, 0], ids=["inline", "mapper_thread"])
@pytest.mark.parametrize("fail_after", [0, 3, 5, 9])
def test_failed_map_is_rolled_back(flags, fail_after):
    """A hipMemMap / hipMemCreate that fails mid-batch (the reference: exit(1), cudaInternal.h:1-13) must leave bookkeeping equal
    to the driver state: the page-groups that did not get all their pages are taken back, the error is reported, and the
    manager keeps working."""
    from tests.impls import fake, fake_mapped
    cfg = dict(SHARE_CFG, num_layers=3)
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=flags)
    p.reserve_physical_pages(12 * group)
    s0 = p.alloc_new_batch_idx(64)
    lens = [0] * 4
    lens[s0] = 64
    p.step_async(lens)                                   # one group mapped (+ look-ahead)
    p.pm.wait()
    before = p.pm.state()
    n_before = len(fake_mapped())
    fake().vattn_fake_fail_map_after(fake_counters()["n_map"] + fail_after)
    s1 = p.alloc_new_batch_idx(200)                      # needs 4 groups = 24 map calls
    lens[s1] = 200
    with pytest.raises(RuntimeError, match="hipMemMap failed"):
        p.step_async(lens)
    fake().vattn_fake_fail_map_after(1 << 62)
    st = p.pm.state()
    whole_groups = fail_after // 6                       # groups that were complete before the failing call
    assert st["mapped"][s1] == whole_groups and st["mapped"][s0] == before["mapped"][s0]
    assert st["pool"] == before["pool"] - whole_groups * 6
    assert len(st["pool_ids"]) == len(set(st["pool_ids"]))
    assert len(fake_mapped()) == n_before + whole_groups * 6          # driver state == bookkeeping
    assert p.pm.stats()["rollbacks"] == 1
    p.step_async(lens)                                   # the same request now goes through
    p.pm.wait()
    assert p.pm.state()["mapped"][s1] >= 4
    assert p.mapped_ranges() is not None and fake_counters()["violations"] == 0
    p.pm.cleanup(); p.pm.close()
    c = fake_counters()
    assert c["mapped_pages"] == 0 and c["live_handles"] == 0


def test_background_map_failure_is_reported_once_and_rolled_back():
    from tests.impls import fake, fake_mapped
    cfg = SHARE_CFG
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=0)
    p.reserve_physical_pages(12 * group)
    s = p.alloc_new_batch_idx(63)
    lens = [0] * 4
    lens[s] = 63
    fake().vattn_fake_fail_map_after(fake_counters()["n_map"] + 4 + 1)   # the sync group (4 maps) passes, the look-ahead fails
    p.step_async(lens)            # the look-ahead (tokens 64.. need a second group) is planned for the mapper
    with pytest.raises(RuntimeError, match="hipMemMap failed"):
        p.pm.wait()
    fake().vattn_fake_fail_map_after(1 << 62)
    st = p.pm.state()
    assert st["mapped"][s] == 1 and len(fake_mapped()) == 4 and st["pool"] == 12 * 4 - 4
    p.pm.wait()                   # reported once
    p.step_async(lens)
    p.pm.wait()
    assert p.pm.state()["mapped"][s] == 2
    p.pm.cleanup(); p.pm.close()


def test_slot_fences_replace_the_device_wide_quiesce():
    """Unmapping a FREED slot's pages must wait for the kernels launched before the free.  With a fence recorded by
    free_batch_idx_on_stream only that point is waited for; a plain free falls back to one device-wide quiesce per batch; pages
    an ACTIVE slot merely no longer needs are unmapped without any wait."""
    from tests.impls import fake
    f = fake()
    f.vattn_fake_quiesce_count.restype = f.vattn_fake_fence_wait_count.restype = __import__("ctypes").c_uint64
    cfg = SHARE_CFG
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=4)
    p.reserve_physical_pages(8 * group)
    a = p.alloc_new_batch_idx(300); b = p.alloc_new_batch_idx(100)
    lens = [0] * 4
    lens[a], lens[b] = 300, 100
    p.step(lens, True)
    q0, w0 = f.vattn_fake_quiesce_count(), f.vattn_fake_fence_wait_count()
    p.pm.free_batch_idx(a, stream=0x1234)               # engine frees slot a on its compute stream
    lens[a] = 0
    p.step(lens, True)                                  # eager reclaim of slot a: fence wait, no quiesce
    assert f.vattn_fake_fence_wait_count() == w0 + 1 and f.vattn_fake_quiesce_count() == q0
    assert p.pm.stats()["fence_waits"] == 1 and p.pm.stats()["quiesce_calls"] == 0
    p.pm.free_batch_idx(b)                              # plain free: no fence -> quiesce fallback
    lens[b] = 0
    p.step(lens, True)
    assert f.vattn_fake_quiesce_count() == q0 + 1
    # an active slot shrunk by on-demand reclaim: no wait at all
    c_ = p.alloc_new_batch_idx(500)
    lens[c_] = 500
    p.step(lens, False)
    lens[c_] = 100                                      # (a restarted request) 8 groups mapped, 2 needed
    d = p.alloc_new_batch_idx(300)
    lens[d] = 300                                       # needs 5 groups, the pool has none left: reclaim from the active slot c_
    q1, w1 = f.vattn_fake_quiesce_count(), f.vattn_fake_fence_wait_count()
    p.step(lens, False)
    assert p.pm.state()["mapped"][c_] == 2 and p.pm.state()["mapped"][d] == 5
    assert f.vattn_fake_quiesce_count() == q1 and f.vattn_fake_fence_wait_count() == w1
    assert fake_counters()["violations"] == 0 and fake_counters()["stale_vas"] == 0
    p.pm.cleanup(); p.pm.close()


def test_inherited_pages_of_a_reactivated_slot_are_unmapped_behind_the_fence():
    """A slot freed right after LAUNCHING its last iteration (free_batch_idx_on_stream) can be handed to a new, shorter request by
    the next step; the pages it inherits beyond the new request's need may still be read by the previous occupant's kernel.  An
    on-demand reclaim that shrinks the now ACTIVE slot must wait for the slot's fence before unmapping them (it used to unmap at
    once: the active-slot shortcut), while pages mapped for the current occupant still go without any wait."""
    from tests.impls import fake
    f = fake()
    cfg = SHARE_CFG                                     # 64 tokens per page-group
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=4)
    p.reserve_physical_pages(8 * group)
    a = p.alloc_new_batch_idx(500)                      # 8 groups: the whole pool
    lens = [0] * 4
    lens[a] = 500
    p.step(lens, False)
    p.pm.free_batch_idx(a, stream=0x1234)               # freed behind the launch of its last iteration
    lens[a] = 0
    a2 = p.alloc_new_batch_idx(100)                     # best fit: the same slot, 8 groups inherited, 2 needed
    assert a2 == a
    lens[a2] = 100
    b = p.alloc_new_batch_idx(200)                      # 4 groups, pool empty: reclaim from the active slot a2
    lens[b] = 200
    q0, w0 = f.vattn_fake_quiesce_count(), f.vattn_fake_fence_wait_count()
    p.step(lens, False)
    st = p.pm.state()
    assert st["mapped"][a2] == 2 and st["mapped"][b] == 4      # (reclaim shrinks a victim to what it needs, vattention.cu:420-438)
    assert f.vattn_fake_fence_wait_count() == w0 + 1 and f.vattn_fake_quiesce_count() == q0
    # the slot grows again (pages of the CURRENT occupant), is then cut back by a restart: those pages go without a wait,
    # the remaining inherited ones (positions 0, 1) still wait
    p.pm.free_batch_idx(b, stream=0x1234); lens[b] = 0
    p.step(lens, True)                                  # eager reclaim returns b's 4 groups (fence wait for slot b)
    lens[a2] = 450                                      # 8 groups: positions 2..7 are new
    p.step(lens, False)
    lens[a2] = 260                                      # 5 needed
    c_ = p.alloc_new_batch_idx(190)
    lens[c_] = 190                                      # 3 groups, pool empty: a2 gives up positions 7, 6, 5 — all its own
    q1, w1 = f.vattn_fake_quiesce_count(), f.vattn_fake_fence_wait_count()
    p.step(lens, False)
    assert p.pm.state()["mapped"][a2] == 5 and p.pm.state()["mapped"][c_] == 3
    assert f.vattn_fake_quiesce_count() == q1 and f.vattn_fake_fence_wait_count() == w1
    lens[a2] = 64                                       # 1 needed: positions 4, 3, 2 (own) and 1 (inherited) go
    d = p.alloc_new_batch_idx(250)
    lens[d] = 250
    p.step(lens, False)
    assert p.pm.state()["mapped"][a2] == 1 and p.pm.state()["mapped"][d] == 4
    assert f.vattn_fake_fence_wait_count() == w1 + 1 and f.vattn_fake_quiesce_count() == q1
    assert fake_counters()["violations"] == 0 and fake_counters()["stale_vas"] == 0
    p.pm.cleanup(); p.pm.close()


@pytest.mark.parametrize("tokens", [1000, 1020], ids=["no_lookahead_batch", "lookahead_batch_queued_behind"])
def test_layer_ordered_mapper_failure_takes_the_whole_groups_back(tokens):
    """VATTN_FLAG_LAYERED_ASYNC: the first layers of a new prompt's page-groups are mapped by step_async itself, the rest by the
    mapper.  When the MAPPER's half fails (transient hipMemMap / hipMemCreate error), the failed groups must be rolled back in ALL
    layers — also the synchronously mapped ones — or they stay mapped under a slot whose count no longer covers them and the next
    grow at that position maps over them: the slot would be unusable for good.
    1020 tokens: step_async also queues a LOOK-AHEAD batch (token 1025 needs a 17th group) behind the layer-ordered one; its maps lie
    above the positions that are taken back, so it must not run — and must be reverted — once the batch ahead of it has failed
    (round 3 left it mapped: ADVICE r3)."""
    from tests.impls import fake, fake_mapped
    from vattention_amd import _lib as L
    cfg = dict(num_layers=8, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=4096, itemsize=2,
               page_size=32 << 10, megacache=False)          # 64 tokens per page
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=L.FLAG_LAYERED_ASYNC)
    p.pm.set_sync_layers(2)
    p.reserve_physical_pages(40 * group)
    s = p.alloc_new_batch_idx(tokens)                    # 16 groups: 64 synchronous maps (2 layers), 192 on the mapper
    lens = [0] * 4
    lens[s] = tokens
    fake().vattn_fake_fail_map_after(fake_counters()["n_map"] + 64 + 100)    # fails inside layer 5 of the mapper's half
    p.step_async(lens)
    with pytest.raises(RuntimeError, match="hipMemMap failed"):
        p.pm.wait()
    fake().vattn_fake_fail_map_after(1 << 62)
    st = p.pm.state()
    # layer-sorted execution: every group misses its later layers, so every group of the step goes back — in every layer
    assert st["mapped"][s] == 0 and len(fake_mapped()) == 0 and st["pool"] == 40 * 16 and st["pagemap_rows"] == 0
    assert sorted(st["pool_ids"]) == list(range(40 * 16))
    assert p.pm.stats()["rollbacks"] == (1 if tokens == 1000 else 2)
    assert p.pm.stats()["quiesce_calls"] >= 1            # the synchronously mapped layers may be in use: the device drains before they are unmapped
    p.step_async(lens)                                   # and the slot is usable again
    p.pm.wait()
    assert p.pm.state()["mapped"][s] == (16 if tokens == 1000 else 17) and len(fake_mapped()) >= 16 * 16
    assert fake_counters()["violations"] == 0
    p.pm.cleanup(); p.pm.close()


def test_layer_ordered_async_mapping():
    """VATTN_FLAG_LAYERED_ASYNC: step_async returns once layers [0, sync_layers) of a new prompt's pages are mapped; the mapper
    maps the remaining layers in order and wait_layer(l) gates layer l.  End state identical to the plain path."""
    import threading
    from vattention_amd import _lib as L
    cfg = dict(num_layers=8, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=4096, itemsize=2,
               page_size=32 << 10, megacache=False)          # 64 tokens per page
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    ref = ProductImpl(cfg, flags=0)
    ref.reserve_physical_pages(40 * group)
    s = ref.alloc_new_batch_idx(1000)
    lens = [0] * 4
    lens[s] = 1000
    ref.step_async(lens); ref.pm.wait()
    want_state, want_ranges = ref.pm.state(), ref.mapped_ranges()
    want_maps = ref.pm.stats()["map_calls"]
    ref.pm.cleanup(); ref.pm.close()

    p = ProductImpl(cfg, flags=L.FLAG_LAYERED_ASYNC)
    p.pm.set_sync_layers(2)
    p.reserve_physical_pages(40 * group)
    assert p.alloc_new_batch_idx(1000) == s
    p.step_async(lens)
    st = p.pm.stats()
    assert st["layered_batches"] == 1
    seen = []
    for layer in range(cfg["num_layers"]):
        p.pm.wait_layer(layer)
        assert p.pm.layers_ready() > layer
        seen.append(p.pm.layers_ready())
    assert seen == sorted(seen)
    p.pm.wait()
    assert p.pm.layers_ready() == cfg["num_layers"]
    assert p.pm.state() == want_state and p.mapped_ranges() == want_ranges
    st = p.pm.stats()
    assert st["map_calls"] == want_maps
    # the synchronous share is sync_layers / num_layers of the prompt's maps (16 groups x 2 layers x 2 tensors)
    assert st["sync_batches"] == 1 and st["async_batches"] >= 1
    assert fake_counters()["violations"] == 0
    # a second, small step (one page-group) stays on the plain path
    lens[s] = 1030
    p.step_async(lens); p.pm.wait()
    assert p.pm.stats()["layered_batches"] == 1
    p.pm.cleanup(); p.pm.close()


def test_admission_lookahead_premap():
    """premap(seqlen): the slot the next request will get is reserved and its pages are mapped by the mapper thread; the step that
    activates the request maps nothing synchronously; alloc_new_batch_idx never hands the reserved slot to anyone else; the end
    state equals the plain alloc + step path (same pages at the same offsets)."""
    cfg = dict(num_layers=4, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=4096, itemsize=2,
               page_size=32 << 10, megacache=False)          # 64 tokens per page
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    ref = ProductImpl(cfg, flags=0)
    ref.reserve_physical_pages(60 * group)
    a = ref.alloc_new_batch_idx(700)
    lens = [0] * 4
    lens[a] = 700
    ref.step_async(lens); ref.pm.wait()
    b = ref.alloc_new_batch_idx(1500)
    lens[b] = 1500
    ref.step_async(lens); ref.pm.wait()
    want_state, want_ranges = ref.pm.state(), ref.mapped_ranges()
    ref.pm.cleanup(); ref.pm.close()

    p = ProductImpl(cfg, flags=0)
    p.reserve_physical_pages(60 * group)
    assert p.alloc_new_batch_idx(700) == a
    lens = [0] * 4
    lens[a] = 700
    p.step_async(lens)
    slot = p.pm.premap(1500)                          # while request `a`'s iteration "runs"
    assert slot == b
    other = p.alloc_new_batch_idx(64)                 # a third request arriving meanwhile must not get the reserved slot
    assert other not in (a, slot)
    p.free_batch_idx(other)
    p.pm.wait()
    st0 = p.pm.stats()
    assert p.pm.state()["mapped"][slot] == 24 and p.pm.state()["lens"][slot] == 0
    lens[slot] = 1500                                 # the engine activates the request
    p.step_async(lens)
    st1 = p.pm.stats()
    assert st1["sync_batches"] == st0["sync_batches"], "the activating step mapped on the critical path"
    p.pm.wait()
    assert p.pm.state() == want_state and p.mapped_ranges() == want_ranges
    # a cancelled look-ahead leaves reclaimable pages and frees the slot for the next allocation
    s3 = p.pm.premap(600)
    assert s3 not in (a, slot) and s3 >= 0
    p.pm.wait()
    assert p.pm.state()["mapped"][s3] == 10
    p.pm.cancel_premap(s3)
    assert p.alloc_new_batch_idx(600) == s3           # reused, pages already in place
    # no free slot -> -1, nothing reserved
    lens[s3] = 600
    p.step_async(lens); p.pm.wait()
    s4 = p.pm.premap(100)
    assert s4 >= 0
    assert p.pm.premap(100) == -1
    # a look-ahead never reclaims: with the pool exhausted it only reserves
    p.pm.cancel_premap(s4)
    assert fake_counters()["violations"] == 0 and fake_counters()["stale_vas"] == 0
    p.pm.cleanup(); p.pm.close()


def test_premapped_pages_are_the_last_to_be_reclaimed():
    cfg = dict(num_layers=2, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=4096, itemsize=2,
               page_size=32 << 10, megacache=False)
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=0)
    p.reserve_physical_pages(30 * group)
    a = p.alloc_new_batch_idx(640)                    # 10 groups
    lens = [0] * 4
    lens[a] = 640
    p.step_async(lens); p.pm.wait()
    p.free_batch_idx(a)                               # deferred reclamation: 10 cached groups on an inactive slot
    lens[a] = 0
    r = p.pm.premap(640)                              # takes the cached slot (smallest sufficient): no new maps
    assert r == a
    o = p.alloc_new_batch_idx(1280)                   # 20 groups: exactly what is left in the pool
    assert o != r
    lens[o] = 1280
    p.step_async(lens); p.pm.wait()
    # the pool is dry now; the step's own look-ahead (one more page) could only come from the reserved slot, and takes exactly
    # what it needs (the reference's reclaim would strip the whole inactive slot, vattention.cu:420-438)
    m0 = p.pm.state()["mapped"]
    assert m0[o] + m0[r] == 30 and m0[r] >= 8
    lens[o] = 1400
    p.step_async(lens); p.pm.wait()
    st = p.pm.state()
    assert st["mapped"][o] >= 22 and 0 < st["mapped"][r] < 10 and st["mapped"][o] + st["mapped"][r] == 30
    assert fake_counters()["violations"] == 0
    p.pm.cleanup(); p.pm.close()


def test_premap_reclaims_from_finished_slots_in_the_background():
    """Pool dry at look-ahead time: premap takes the pages from slots that hold more than they need on the MAPPER thread (the
    unmaps, and the wait for those slots' fences, run there), so neither the look-ahead nor the step that activates the request
    executes driver calls on the caller's thread."""
    cfg = dict(num_layers=2, num_kv_heads=2, head_size=128, max_batch_size=4, max_context_length=4096, itemsize=2,
               page_size=32 << 10, megacache=False)          # 64 tokens per page
    group = 2 * cfg["num_layers"] * cfg["page_size"]
    p = ProductImpl(cfg, flags=0)
    p.reserve_physical_pages(19 * group)
    a = p.alloc_new_batch_idx(576)                    # 9 groups
    b = p.alloc_new_batch_idx(640)                    # 10 groups
    lens = [0] * 4
    lens[a], lens[b] = 576, 640
    p.step_async(lens); p.pm.wait()
    p.pm.free_batch_idx(a); p.pm.free_batch_idx(b)    # both finished: their pages stay mapped (deferred reclamation)
    lens[a] = lens[b] = 0
    c = p.alloc_new_batch_idx(30)                     # a short request reuses one of them and keeps its surplus pages for now
    lens[c] = 30
    p.step_async(lens); p.pm.wait()
    m0 = p.pm.state()["mapped"]
    assert sum(m0) == 19 and p.pm.state()["pool"] == 0, "the pool must be dry for this scenario"
    st0 = p.pm.stats()
    s = p.pm.premap(1152)                             # 18 groups: 10 are there, the other 8 come out of slot c's surplus
    assert s >= 0 and s != c
    p.pm.wait()
    st1 = p.pm.stats()
    assert st1["sync_batches"] == st0["sync_batches"], "a look-ahead executed driver calls on the caller's thread"
    assert st1["unmap_calls"] > st0["unmap_calls"] and p.pm.state()["mapped"][s] == 18
    lens[s] = 1152                                    # activation: everything is in place
    p.step_async(lens)
    assert p.pm.stats()["sync_batches"] == st1["sync_batches"]
    p.pm.wait()
    assert p.pm.state()["mapped"][c] >= 1
    assert fake_counters()["violations"] == 0 and fake_counters()["stale_vas"] == 0
    p.pm.cleanup(); p.pm.close()
