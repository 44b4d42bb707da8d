#!/usr/bin/env python3
"""LAB: runs tools/power_ceiling_probe (whole-chip MFMA streams, ~1 s each) under the clock / power sampler of the bench
(vattention_amd/telemetry.py) and prints every line with the clock and board power of its second half.  usage: python tools/lab/power_ceiling.py [--operand-reuse]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vattention_amd.telemetry import Sampler      # noqa: E402

from vattention_amd import build as B      # noqa: E402
exe = B.build_probe()
with Sampler(0, interval=0.02) as s:
    out = subprocess.run([exe] + sys.argv[1:], stdout=subprocess.PIPE, text=True, timeout=300).stdout
    s.stop()
    for line in out.splitlines():
        m = re.search(r" t0=([\d.]+) t1=([\d.]+)$", line)
        if not m:
            print(line)
            continue
        w = s.window(float(m.group(1)), float(m.group(2)))
        print("%s   clock %s MHz (min %s), power %s W  [%d samples, %s]" % (line[:m.start()], w.get("clock_mhz_mean"), w.get("clock_mhz_min"), w.get("power_w_mean"), w.get("samples", 0), w.get("source")))
