"""The launch plans, pinned WITHOUT a stopwatch.  vattn_attn_plan_describe (include/vattn_kernels.h) is pure host
arithmetic: for every shape of the timing gate (tools/plan_gate.py, which is where the wall-clock comparison lives now — its output goes
to profiles/) and for the shapes BASELINE.json's configs launch, the plan must equal the committed table; and the plan must be EXACTLY
an explicit (tiling, shares) launch of the product library — "the default and the explicit tiling it chose are the same launch" was
assumed by the timing gate and asserted nowhere.  The reference's equivalents are the launch heuristics of
/root/reference/pod_attn/pod_attn/flash_api.cpp:258-323 and flash_fwd_launch_template.h:100-162.  A heuristic change shows up HERE, as a
diff of this table, not as a flaky timing assertion on a slow box."""
import pytest

from vattention_amd import kernels as K


def _params(b, sq, sk, h, hk, d=128, hint=0, causal=1, splits=0, variant=0, knew=0):
    p = K.AttnParams()
    p.b, p.seqlen_q, p.seqlen_k, p.seqlen_knew, p.h, p.h_k, p.d = b, sq, sk, knew, h, hk, d
    p.is_causal, p.dtype, p.num_splits, p.variant, p.max_seqlen_k_hint = causal, 0, splits, variant, hint
    return p


# (name, query heads, kv heads, chunk tokens n, cached tokens c) -> (path, tiling, shares, workgroups, merge launch, workspace bytes)
# path 0 = grid order; tiling 1 = 8 waves x 32 rows, 4 = 4 waves x 32 rows, 7 = prefill64
PREFILL = {
    ("small 2k (32/4 heads)", 32, 4, 2048, 0): (0, 4, 1, 512, 0, 0),
    ("llama70b/tp8 2k", 8, 1, 2048, 0): (0, 4, 2, 256, 1, 16908288),
    ("llama70b/tp8 4k", 8, 1, 4096, 0): (0, 4, 4, 1024, 1, 67633152),
    ("llama70b/tp8 8k", 8, 1, 8192, 0): (0, 1, 2, 512, 1, 67633152),
    ("yi6b chunk4k@0", 32, 4, 4096, 0): (0, 7, 1, 512, 0, 0),
    ("llama8b chunk512@8k", 32, 8, 512, 7680): (0, 7, 4, 256, 1, 33816576),
    ("llama70b/tp8 chunk512@16k", 8, 1, 512, 15872): (0, 4, 8, 256, 1, 16908288),
    ("llama70b/tp8 chunk2k@30k", 8, 1, 2048, 30720): (0, 7, 4, 256, 1, 33816576),
    ("configs[1] yi6b whole prompt", 32, 4, 32702, 0): (0, 7, 1, 4096, 0, 0),
    ("configs[1] yi6b chunk4k@28k", 32, 4, 4096, 28672): (0, 7, 1, 512, 0, 0),
    ("configs[3] yi34b/tp2 chunk16k@112k", 28, 4, 16384, 114688): (0, 7, 1, 1792, 0, 0),
    ("configs[3] yi34b/tp4 chunk16k@112k", 14, 2, 16384, 114688): (0, 7, 2, 1792, 1, 236716032),
    ("configs[4] llama70b/tp8 longest prompt", 8, 1, 29092, 0): (0, 7, 1, 912, 0, 0),
    ("configs[2] llama8b 16k prompt", 32, 8, 16384, 0): (0, 7, 1, 2048, 0, 0),
}
GATE_SHAPES = [k for k in PREFILL if not k[0].startswith("configs")]

# (name, query heads, kv heads, batch, context) -> the same tuple.  path 0 = uniform split of every sequence (grid heuristics), 2 = the
# device-planned stream decomposition (no host lengths); tiling = 16-head blocks per workgroup
DECODE = {
    ("configs[1] yi6b B16@32k", 32, 4, 16, 32768): (2, 1, 0, 768, 1, 6922368),
    ("yi6b B1@32k", 32, 4, 1, 32768): (0, 1, 48, 192, 1, 792576),
    ("yi6b B4@32k", 32, 4, 4, 32768): (2, 1, 0, 768, 1, 6523008),
    ("llama8b B64@8k", 32, 8, 64, 8192): (2, 1, 0, 768, 1, 10650112),
    ("configs[2] llama8b B256@2k", 32, 8, 256, 2048): (2, 1, 0, 768, 1, 23431168),
    ("configs[4] llama70b/tp8 B64@32k", 8, 1, 64, 32768): (2, 1, 0, 768, 1, 6922752),
    ("configs[4] llama70b/tp8 B256@32k", 8, 1, 256, 32768): (2, 1, 0, 768, 1, 8521728),
    ("configs[3] yi34b/tp2 B8@128k", 28, 4, 8, 131072): (2, 1, 0, 768, 1, 6656128),
    ("configs[3] yi34b/tp2 B1@128k", 28, 4, 1, 131072): (0, 1, 64, 256, 1, 924672),
    ("yi34b/tp4 B1@128k", 14, 2, 1, 131072): (0, 1, 96, 192, 1, 693504),
    ("mqa G32 B16@16k", 32, 1, 16, 16384): (0, 2, 32, 512, 1, 8454144),
    ("mqa G64 B16@8k", 64, 1, 16, 8192): (0, 2, 16, 512, 1, 8454144),
    ("yi6b B16@2k", 32, 4, 16, 2048): (2, 1, 0, 768, 1, 6922368),
    ("falcon G71 d64 B8@4k", 71, 1, 8, 4096): (0, 2, 32, 768, 1, 4725760),
}


def _tuple(d):
    return (d["path"], d["tiling"], d["nsplit"], d["workgroups"], d["merge_launch"], d["workspace_bytes"])


@pytest.mark.parametrize("key", list(PREFILL), ids=[k[0] for k in PREFILL])
def test_prefill_plan_table(key):
    _name, Hq, Hkv, n, c = key
    d = K.describe(_params(1, n, c + n, Hq, Hkv, hint=c + n))
    assert d["form"] == 0 and _tuple(d) == PREFILL[key], (key, d)


@pytest.mark.parametrize("key", GATE_SHAPES, ids=[k[0] for k in GATE_SHAPES])
def test_default_plan_is_exactly_an_explicit_launch(key):
    """The plan picks among the product's tilings and share counts; asking for that tiling and that share count explicitly
    (variant bits 1-3, num_splits) must describe the SAME launch — what tools/plan_gate.py times against each other."""
    _name, Hq, Hkv, n, c = key
    d = K.describe(_params(1, n, c + n, Hq, Hkv, hint=c + n))
    e = K.describe(_params(1, n, c + n, Hq, Hkv, hint=c + n, variant=d["tiling"] << 1, splits=d["nsplit"]))
    assert _tuple(e) == _tuple(d), (key, d, e)


@pytest.mark.parametrize("key", list(DECODE), ids=[k[0] for k in DECODE])
def test_decode_plan_table(key):
    name, Hq, Hkv, B, ctx = key
    d = K.describe(_params(B, 1, ctx, Hq, Hkv, d=64 if "d64" in name else 128, knew=1))
    assert d["form"] == 1 and _tuple(d) == DECODE[key], (key, d)


def test_decode_plan_needs_no_host_lengths_and_the_grid_heuristics_stay_selectable():
    """Batches of two or more sequences take the device-planned stream decomposition whatever the caller knows about the lengths; variant
    bit 19 selects the grid heuristics of rounds 1-3 for A/B; explicit split counts keep meaning what they meant."""
    p = _params(256, 1, 32768, 8, 1, knew=1)
    assert K.describe(p)["path"] == 2
    p.variant = K.LEGACY_DECODE_PLAN
    d = K.describe(p)
    assert d["path"] == 0 and d["nsplit"] == 3 and d["workgroups"] == 768
    assert K.describe(_params(16, 1, 32768, 32, 4, knew=1, splits=5))["nsplit"] == 5
    assert K.describe(_params(300, 1, 2048, 8, 2, knew=1))["path"] == 0          # beyond the 256 sequences the plan prologue takes
    f = K.describe(_params(16, 1, 32768, 32, 4, knew=1, splits=-100))            # forced workgroup count (tests, A/B)
    assert f["path"] == 2 and f["workgroups"] == 400
