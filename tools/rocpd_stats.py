#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` SQLite result (ROCm 7.2 writes rocpd .db files) as the
per-kernel table the judge reads: calls, total ms, mean/min/max us, % of GPU kernel time, grid, regs.
usage: tools/rocpd_stats.py <results.db> [> profiles/rNN_<what>_kernel_stats.md]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(grid_z), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | mean us | min us | max us | % | grid | wg | vgpr | agpr | sgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0]
        if len(name) > 90:
            name = name[:87] + "..."
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f | %dx%dx%d | %d | %d | %d | %d | %d |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13]))


if __name__ == "__main__":
    main(sys.argv[1])
