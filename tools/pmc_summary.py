#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.  ROCm 7.2 writes rocpd SQLite databases (older setups: *counter_collection.csv); both are
read.  usage: tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub in k:
            acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k[:60]][r["Counter_Name"]] += 1
for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    # one row per (dispatch, counter instance): sum the instances of a dispatch first, then average over dispatches
    q = ("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? "
         "group by kernel_name, counter_name, dispatch_id")
    try:
        rows = db.execute(q, ("%" + sub + "%",)).fetchall()
    except sqlite3.OperationalError:
        rows = [(r[0], r[1], i, r[2]) for i, r in enumerate(db.execute(
            "select kernel_name, counter_name, value from counters_collection where kernel_name like ?", ("%" + sub + "%",)))]
    for k, c, _disp, v in rows:
        acc[k[:60]][c] += float(v)
        cnt[k[:60]][c] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-32s total %.4g  per-dispatch %.4g  (n=%d)" % (c, acc[k][c], acc[k][c] / cnt[k][c], cnt[k][c]))
