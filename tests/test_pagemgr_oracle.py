"""The Python oracle (oracle/pagemgr.py) against the REAL reference's answers.

(1) golden: traces + answers recorded from oracle/_ref (reference vattention.cu compiled against a
    fake CUDA driver) and committed under tests/golden/ — runs everywhere;
(2) live: fresh random traces against oracle/_ref where it is built (build container only).
Bit-exact: return values, error text, mapped_pages[], curr_seq_lengths[], pool size AND order,
page map, and the driver-call log.
"""
import os

import pytest

from oracle import trace as T
from tests.golden_util import load, pagemgr_files

FILES = pagemgr_files()


def test_golden_present():
    assert len(FILES) >= 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[8:-8] for p in FILES])
def test_oracle_matches_reference_golden(path):
    g = load(path)
    cfg = g["config"]
    nops = 0
    for tr in g["traces"]:
        got = T.normalize_ops(T.replay(T.OracleImpl(cfg), tr["ops"], full=True), cfg["page_size"])
        exp = tr["expect"]
        assert len(got) == len(exp)
        for i, (a, b) in enumerate(zip(got, exp)):
            assert a == b, "trace %s/%s op %d %s" % (tr["kind"], tr["seed"], i, tr["ops"][i][:1])
        nops += len(got)
    assert nops > 1000


def test_tensor_shapes_match_reference():
    from oracle.pagemgr import PageManagerOracle
    for path in FILES:
        g = load(path)
        c = g["config"]
        o = PageManagerOracle(c["num_layers"], c["num_kv_heads"], c["head_size"], c["max_batch_size"],
                              c["max_context_length"], c["itemsize"], c["page_size"], c["megacache"])
        assert list(o.tensor_shape()) == g["tensor_info"]["shape"]
        assert o.num_tensors == g["tensor_info"]["n"]


def test_survey_a3_arithmetic():
    """SURVEY §A.3 / BASELINE.md §3 per-config constants (fp16, D=128)."""
    from oracle.pagemgr import PageManagerOracle, MB, KB
    rows = [  # L, kvh, page, ctx, B -> tokens/page, pages/req, virt/req
        (32, 4, 2 * MB, 32768, 16, 2048, 16, 32 * MB),
        (32, 8, 64 * KB, 32768, 256, 32, 1024, 64 * MB),
        (60, 4, 2 * MB, 131072, 50, 2048, 64, 128 * MB),
        (80, 1, 256 * KB, 32768, 256, 1024, 32, 8 * MB),
    ]
    for L, kvh, page, ctx, B, tpp, ppr, vpr in rows:
        o = PageManagerOracle(L, kvh, 128, B, ctx, 2, page, False)
        assert (o.tokens_per_page, o.max_pages_per_req, o.virt_buff_size_per_req) == (tpp, ppr, vpr)
        # 0.9 x 288 GB pool
        n = o.reserve_physical_pages(int(288e9 * 0.9)) if page == 2 * MB else None
        if n is not None:
            assert n // (2 * L) in (1931, 1029)


def test_u64_wraparound_is_kept():
    from oracle.pagemgr import PageManagerOracle, MB
    o = PageManagerOracle(2, 8, 128, 4, 16384, 2, 2 * MB, False)
    o.reserve_physical_pages(64 * MB)
    r = o.alloc_new_batch_idx(5000)      # length set before anything is mapped -> negative term wraps
    assert r == 0
    assert o.num_free_kvblocks() == (8 - 5) % (1 << 64)
    o2 = PageManagerOracle(2, 8, 128, 4, 16384, 2, 2 * MB, False)
    o2.alloc_new_batch_idx(5000)
    assert o2.num_free_kvblocks() == (1 << 64) - 5


def test_live_reference_random_traces():
    ref = pytest.importorskip("oracle.ref_adapter")
    if not ref.available():
        pytest.skip("oracle/_ref not built here")
    cfgs = [dict(num_layers=3, num_kv_heads=8, head_size=128, max_batch_size=7, max_context_length=12288,
                 itemsize=2, page_size=2 << 20, megacache=False),
            dict(num_layers=2, num_kv_heads=4, head_size=128, max_batch_size=9, max_context_length=3072,
                 itemsize=2, page_size=128 << 10, megacache=False)]
    for ci, cfg in enumerate(cfgs):
        for seed in range(4):
            if seed < 3:
                tr = T.gen_serving_trace(cfg, 900 + seed, iters=80, pool_groups=[30, 10, 18][seed], use_async=seed != 1,
                                         chunk=[0, 1024, 300][seed], p_finish=0.04)
                ops = T.resolve(tr, T.OracleImpl)
            else:
                ops = T.gen_adversarial_trace(cfg, 950, 120, 9)["ops"]
            a = T.replay(T.OracleImpl(cfg), ops, full=True)
            ops = T.truncate_for_reference(ops, a)
            a = T.normalize_ops(T.replay(T.OracleImpl(cfg), ops, full=True), cfg["page_size"])
            b = T.replay(ref.RefImpl(cfg), ops, full=True)
            assert a == b, (ci, seed)
