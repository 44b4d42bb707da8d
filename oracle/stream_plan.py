"""CPU restatement of the decode launch's DEVICE-side plan (vattention_amd/csrc/decode_body.h: stream_plan_load, stream_geom,
stream_seq_records, decode_stream_kernel).  TEST INFRASTRUCTURE ONLY (tests/ import it; the product never does).

The reference has no counterpart to pin this against — its split heuristic gives every sequence the same number of splits
(/root/reference/pod_attn/pod_attn/flash_api.cpp:258-323) — so the restatement is checked two ways: its invariants hold for arbitrary
batches (tests/test_stream_plan_model.py: every 32-key tile of every sequence is owned by exactly one workgroup, a sequence's pieces
are contiguous, ordered and published as consecutive, globally unique records), and the kernel's own plan (the (first record, count)
table the launch leaves in its workspace) equals it on the GPU (tests/test_gpu_attention.py::test_decode_stream_plan_table_matches_the_model).
"""
from __future__ import annotations

from typing import List, Tuple

DC_BN = 32                 # keys per tile
DC_SWITCH_TILES = 4        # positions every sequence occupies ahead of its first tile
DC_MIN_WG_TILES = 8


def tiles_of(lk: int) -> int:
    return (lk + DC_BN - 1) // DC_BN if lk > 0 else 1          # an empty sequence owns one (masked) tile


def plan(lens: List[int], seqlen_k: int, seqlen_knew: int, nwg: int, switch_tiles: int = DC_SWITCH_TILES):
    """lens = cache_seqlens as the device array holds them.  Returns (uniform?, pieces, records) where pieces[w] = [(b, tile_begin,
    tile_end, record)] in processing order and records[b] = (first record, count)."""
    X = switch_tiles
    lk = [min(max(int(x), 0) + seqlen_knew, seqlen_k) for x in lens]
    t = [tiles_of(x) for x in lk]
    B = len(t)
    incl, acc = [], 0
    for x in t:
        acc += x + X
        incl.append(acc)
    total, maxt = acc, max(t)
    T = max(DC_MIN_WG_TILES, (total + nwg - 1) // nwg)
    S = nwg // B
    uniform = S >= 1 and (maxt + S - 1) // S <= T
    if not uniform:          # a short tile space: fewer, longer ranges (stream_geom)
        tpw = total // nwg
        eff = max(1, nwg // 3) if tpw < 20 else nwg
        T = max(DC_MIN_WG_TILES, (total + eff - 1) // eff)
    pieces: List[List[Tuple[int, int, int, int]]] = [[] for _ in range(nwg)]
    records = []
    if uniform:
        for b in range(B):
            per = (t[b] + S - 1) // S
            cnt = (t[b] + per - 1) // per
            records.append((b * S + b, cnt))
            for s in range(S):
                tb, te = s * per, min(t[b], s * per + per)
                if te > tb:
                    pieces[b * S + s].append((b, tb, te, b * S + s + b))
        return True, pieces, records
    for b in range(B):
        excl = incl[b - 1] if b else 0
        first_w, last_w = (excl + X) // T, (incl[b] - 1) // T
        records.append((first_w + b, last_w - first_w + 1))
    for w in range(nwg):
        g0 = w * T
        if g0 >= total:
            continue
        g1 = min(total, g0 + T)
        for b in range(B):
            excl = incl[b - 1] if b else 0
            if excl >= g1:
                break
            if incl[b] <= g0:
                continue
            real0 = excl + X
            tb, te = max(g0, real0) - real0, min(g1, incl[b]) - real0
            if te > tb:
                pieces[w].append((b, tb, te, w + b))
    return False, pieces, records
