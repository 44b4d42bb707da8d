#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: usage tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k[:60]][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-32s total %.4g  per-dispatch %.4g  (n=%d)" % (c, acc[k][c], acc[k][c] / cnt[k][c], cnt[k][c]))
