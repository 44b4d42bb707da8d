#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --hip-trace rocpd database: how much of the HIP VMM work (hipMemCreate / hipMemMap /
hipMemSetAccess / hipMemUnmap) ran while attention kernels were executing on the GPU, split by host thread (the engine thread
issues the synchronous part of step_async, the mapper thread the look-ahead part).
usage: tools/mapper_overlap.py <results.db>"""
import bisect
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    ks = db.execute("select start, end from kernels order by start").fetchall()
    if not ks:
        print("no kernels in trace")
        return
    # merge kernel intervals into busy intervals
    busy = []
    for s, e in ks:
        if busy and s <= busy[-1][1]:
            busy[-1][1] = max(busy[-1][1], e)
        else:
            busy.append([s, e])
    starts = [b[0] for b in busy]

    def overlap(s, e):
        i = max(0, bisect.bisect_right(starts, s) - 1)
        tot = 0
        while i < len(busy) and busy[i][0] < e:
            tot += max(0, min(e, busy[i][1]) - max(s, busy[i][0]))
            i += 1
        return tot

    cols = [r[1] for r in db.execute("pragma table_info(regions)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = db.execute("select %s, tid, start, end from regions where %s like 'hipMem%%'" % (name_col, name_col)).fetchall()
    t0, t1 = ks[0][0], max(e for _, e in ks)
    gpu_busy = sum(b[1] - b[0] for b in busy)
    print("trace window %.3f s, GPU busy with kernels %.3f s (%.1f %%), %d kernels" % ((t1 - t0) / 1e9, gpu_busy / 1e9, 100.0 * gpu_busy / (t1 - t0), len(ks)))
    agg = {}
    for name, tid, s, e in rows:
        if name not in ("hipMemCreate", "hipMemMap", "hipMemSetAccess", "hipMemUnmap", "hipMemRelease", "hipMemAddressReserve"):
            continue
        a = agg.setdefault((tid, name), [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        a[2] += overlap(s, e)
    tids = sorted({k[0] for k in agg})
    main_tid = min(tids, key=lambda t: min(r[2] for r in rows if r[1] == t)) if tids else None
    print("| thread | call | count | total ms | ms while kernels ran | overlapped |")
    print("|---|---|---|---|---|---|")
    for (tid, name), (n, dur, ov) in sorted(agg.items()):
        who = "engine thread" if tid == main_tid else "mapper thread"
        print("| %s (%d) | %s | %d | %.2f | %.2f | %.0f %% |" % (who, tid, name, n, dur / 1e6, ov / 1e6, 100.0 * ov / dur if dur else 0))


if __name__ == "__main__":
    main(sys.argv[1])
