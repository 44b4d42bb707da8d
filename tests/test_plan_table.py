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


# ---- persistent work lists (vattn_prefill_plan_wg, round 5): pure host arithmetic ----
def _plan_wg(b, q_lens, k_lens, h, h_k, max_wg=0, force_tiles=0):
    import ctypes as C
    from vattention_amd import kernels as K
    p = K.AttnParams()
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = b, max(q_lens), h, h_k, 128, 1
    if force_tiles:
        p.num_splits = -force_tiles
    nblk = sum((q + 255) // 256 for q in q_lens) * h
    cap_i, cap_b = 17 * nblk + 16, nblk + 16
    items, blocks = (K.PrefillItem * cap_i)(), (K.PrefillItem * cap_b)()
    counts, wg_first = (C.c_int32 * 4)(), (C.c_int32 * 257)()
    ql, kl = (C.c_int32 * b)(*q_lens), (C.c_int32 * b)(*k_lens)
    n = K.klib().vattn_prefill_plan_wg(C.byref(p), ql, kl, items, cap_i, blocks, cap_b, wg_first, max_wg, counts)
    return n, [items[i] for i in range(max(n, 0))], [blocks[i] for i in range(counts[1])], list(wg_first[:counts[3] + 1]), list(counts)


@pytest.mark.parametrize("case", [
    dict(b=1, q=[9441], k=[9441], h=32, h_k=8),                       # one arxiv-length prompt, llama-3-8b heads: 1 184 blocks, several rounds
    dict(b=1, q=[8192], k=[8192], h=8, h_k=1),                        # the TP8 rank's prompt: one underfilled round, blocks cut
    dict(b=3, q=[23774, 5637, 1000], k=[23774, 5637, 1000], h=8, h_k=1),   # ragged batch
    dict(b=1, q=[2048], k=[32768], h=8, h_k=1),                       # chunk on a long prefix
    dict(b=2, q=[300, 70], k=[300, 70], h=4, h_k=2),                  # tiny: fewer pieces than XCDs
], ids=["llama8b_9k", "tp8_8k", "ragged3", "chunk2k@30k", "tiny"])
def test_persistent_work_list_covers_every_block_once_and_balances_the_queues(case):
    b, q, k, h, h_k = case["b"], case["q"], case["k"], case["h"], case["h_k"]
    n, items, blocks, wg_first, counts = _plan_wg(b, q, k, h, h_k)
    assert n > 0 and counts[0] == n and counts[3] >= 1
    nwg = counts[3]
    assert nwg <= 256 and (nwg % 8 == 0 or nwg == n)
    assert wg_first[0] == 0 and wg_first[-1] == n and all(x <= y for x, y in zip(wg_first, wg_first[1:]))
    assert all(wg_first[w] < wg_first[w + 1] for w in range(nwg)), "an empty queue would be a workgroup without work"
    # every (entry, head, query block) is covered exactly once by contiguous tile ranges that end open
    cover = {}
    for it in items:
        cover.setdefault((it.b, it.h, it.qb), []).append((it.tile_begin, it.tile_end, it.nshares, it.part_row))
    want = {(e, hh, qb) for e in range(b) for hh in range(h) for qb in range((q[e] + 255) // 256)}
    assert set(cover) == want
    for key, pcs in cover.items():
        pcs.sort()
        e, _, qb = key
        tiles = (min(k[e], qb * 256 + 256 + (k[e] - q[e])) + 63) // 64
        assert pcs[0][0] == 0 and pcs[-1][1] == 0x7fffffff and len(pcs) == pcs[0][2]
        for (a0, a1, _, _), (b0, _, _, _) in zip(pcs, pcs[1:]):
            assert a1 == b0 and a0 < a1 <= tiles
        assert (len(pcs) == 1) == (pcs[0][3] == -1)
    # queues: longest first inside a queue; the kv heads of a queue's pieces share its XCD class; loads within a piece of each other
    G = h // h_k
    ncls = h_k if (nwg >= 8 and h_k <= 8 and 8 % h_k == 0) else 1
    length = lambda it, key: (min(it.tile_end, (min(k[key[0]], key[2] * 256 + 256 + (k[key[0]] - q[key[0]])) + 63) // 64) - it.tile_begin)
    loads = []
    for w in range(nwg):
        mine = items[wg_first[w]:wg_first[w + 1]]
        ls = [length(it, (it.b, it.h, it.qb)) for it in mine]
        assert ls == sorted(ls, reverse=True)
        assert all((it.h // G) % ncls == (w % 8) % ncls for it in mine)
        loads.append(sum(2 * x + 2 + (3 if it.nshares > 1 else 0) for x, it in zip(ls, mine)))
    longest = max(2 * length(it, (it.b, it.h, it.qb)) + 5 for it in items)
    for c in range(ncls):
        cl = [loads[w] for w in range(nwg) if (w % 8) % ncls == c]
        assert max(cl) - min(cl) <= longest, "greedy longest-first leaves no queue more than one piece ahead of another"


def test_persistent_work_list_respects_a_workgroup_cap_and_a_forced_piece_length():
    n, items, blocks, wg_first, counts = _plan_wg(1, [8192], [8192], 8, 1, max_wg=64, force_tiles=16)
    assert n > 0 and counts[3] == 64 and wg_first[-1] == n
    assert all((it.tile_end if it.tile_end != 0x7fffffff else it.tile_begin + 16) - it.tile_begin <= 16 for it in items)
    assert counts[1] == len(blocks) and all(bk.nshares > 1 for bk in blocks)


def test_drawn_queues_keep_the_list_longest_first_and_only_choose_the_workgroup_count():
    """vattn_prefill_plan_wg(wg_first_out = NULL): nothing is assigned — the persistent workgroups draw their pieces on the device."""
    import ctypes as C
    from vattention_amd import kernels as K
    p = K.AttnParams()
    q = [6526, 14505, 5364]
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = 3, max(q), 8, 1, 128, 1
    nblk = sum((x + 255) // 256 for x in q) * 8
    items, blocks = (K.PrefillItem * (17 * nblk + 16))(), (K.PrefillItem * (nblk + 16))()
    counts = (C.c_int32 * 4)()
    ql = (C.c_int32 * 3)(*q)
    n = K.klib().vattn_prefill_plan_wg(C.byref(p), ql, ql, items, 17 * nblk + 16, blocks, nblk + 16, None, 0, counts)
    assert n == nblk and counts[3] == 256 and counts[1] == 0          # a balanced ragged batch: compacted, not cut
    lens = [(min(q[items[i].b], items[i].qb * 256 + 256) + 63) // 64 for i in range(n)]
    assert lens == sorted(lens, reverse=True)
    ref = (K.PrefillItem * (17 * nblk + 16))()
    c3 = (C.c_int32 * 3)()
    assert K.klib().vattn_prefill_plan(C.byref(p), ql, ql, ref, 17 * nblk + 16, blocks, nblk + 16, c3) == n
    assert all((items[i].b, items[i].h, items[i].qb) == (ref[i].b, ref[i].h, ref[i].qb) for i in range(n)), "same list as the per-piece launch takes"


def _work_list_digest(Hq, Hkv, q_lens, k_lens, mode, max_wg):
    import ctypes as C
    import hashlib
    from vattention_amd import kernels as K
    p = K.AttnParams()
    B = len(q_lens)
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal, p.seqlen_k = B, max(q_lens), Hq, Hkv, 128, 1, max(k_lens)
    n_blk = sum((q + 255) // 256 for q in q_lens) * Hq
    cap_i, cap_b = 17 * n_blk + 16, n_blk + 16
    items, blocks = (K.PrefillItem * cap_i)(), (K.PrefillItem * cap_b)()
    ql, kl = (C.c_int32 * B)(*q_lens), (C.c_int32 * B)(*k_lens)
    if mode == 0:
        counts = (C.c_int32 * 3)()
        n = K.klib().vattn_prefill_plan(C.byref(p), ql, kl, items, cap_i, blocks, cap_b, counts)
        wf = b""
    else:
        counts = (C.c_int32 * 4)()
        wg = (C.c_int32 * 257)()
        n = K.klib().vattn_prefill_plan_wg(C.byref(p), ql, kl, items, cap_i, blocks, cap_b, wg, max_wg, counts)
        wf = bytes(wg)
    h = hashlib.sha256()
    h.update(bytes(items)[:max(n, 0) * C.sizeof(K.PrefillItem)] + bytes(blocks)[:counts[1] * C.sizeof(K.PrefillItem)] + wf)
    return [n] + list(counts) + [h.hexdigest()[:16]]


def _work_list_cases():
    import random
    rnd = random.Random(20260926)
    out = []
    for _ in range(48):
        Hq, Hkv = rnd.choice([(8, 1), (32, 4), (32, 8), (28, 4), (14, 2)])
        B = rnd.choice([1, 1, 2, 3, 4])
        q_lens = [rnd.randint(300, rnd.choice([5000, 12000, 30000])) for _ in range(B)]
        k_lens = [q + (rnd.randint(0, 100000) if rnd.random() < 0.3 else 0) for q in q_lens]
        mode = rnd.choice([0, 0, 1])
        out.append((Hq, Hkv, q_lens, k_lens, mode, rnd.choice([0, 64, 256]) if mode else 0))
    return out


def test_work_lists_are_the_pinned_ones():
    """The planner's output on 48 seeded launches (one to four prompts, with and without a prefix, per-piece and host-assigned queues) against
    tests/golden/prefill_work_lists.json: the candidate pricing was re-implemented in round 5 (count-per-load replay instead of a sort and a
    heap per candidate — 6 000 random launches compared list for list against the previous build) and must keep choosing the same lists, which
    are the ones the GPU timings of profiles/r05_planner_ab.txt were taken on.  Regenerate with VATTN_REGEN_PLAN_GOLDEN=1 after a DELIBERATE change."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "prefill_work_lists.json")
    got = [_work_list_digest(*c) for c in _work_list_cases()]
    if os.environ.get("VATTN_REGEN_PLAN_GOLDEN") == "1":
        json.dump({"cases": [list(c) for c in _work_list_cases()], "lists": got}, open(path, "w"))
    want = json.load(open(path))["lists"]
    assert got == want
    assert sum(1 for g in got if g[2] > 0) >= 10          # (a fair share of them cut blocks)
