"""Property test: ANY sequence of allocator API calls leaves the product's C++ manager (fake backend) and the Python
oracle in the same observable state, with the same return values / errors, and a consistent physical picture."""
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import trace as T
from tests.impls import ProductImpl, fake_counters

CFG = dict(num_layers=2, num_kv_heads=2, head_size=128, max_batch_size=5, max_context_length=1024, itemsize=2,
           page_size=64 << 10, megacache=False)          # 64 tokens per page, 16 pages per request
GROUP = 2 * CFG["num_layers"] * CFG["page_size"]

lens_st = st.lists(st.integers(0, CFG["max_context_length"]), min_size=5, max_size=5)
op_st = st.one_of(
    st.tuples(st.just("reserve"), st.integers(0, 40).map(lambda g: g * GROUP + 123)),
    st.tuples(st.just("alloc"), st.integers(1, CFG["max_context_length"])),
    st.tuples(st.just("free"), st.integers(0, 4)),
    st.tuples(st.just("step"), lens_st, st.booleans()),
    st.tuples(st.just("step_async"), lens_st),
    st.tuples(st.just("nfree")),
    st.tuples(st.just("set_deferred"), st.booleans()),
    st.tuples(st.just("map_common"), st.integers(0, 200)),
)


MEGA = dict(CFG, megacache=True, num_layers=4)           # 16 tokens per page, 64 pages per request, 2 pages per group


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.lists(op_st, min_size=1, max_size=40), st.sampled_from([0, 4]), st.booleans())
def test_any_call_sequence_matches_oracle(ops, flags, mega):
    ops = [list(o) for o in ops] + [["cleanup"]]
    cfg = MEGA if mega else CFG
    o = T.OracleImpl(cfg, shared_page_refcount=True)     # the product's prefix-sharing fix (see oracle/pagemgr.py)
    p = ProductImpl(cfg, flags=flags)
    try:
        for op in ops:
            ra = T.replay(o, [op], full=True)[0]
            rb = T.replay(p, [op], full=True)[0]
            for k in ("ret", "err", "mapped", "lens", "pool", "pool_handles", "pagemap"):
                assert ra[k] == rb[k], (op, k)
            if op[0] != "map_common" and not any(x[0] == "map_common" for x in ops):
                assert p.mapped_ranges() == o.o.mapped_ranges()
        c = fake_counters()
        assert c["stale_vas"] == 0
        if not any(x[0] == "map_common" for x in ops):
            assert c["violations"] == 0 and c["mapped_pages"] == 0
    finally:
        p.pm.close()
