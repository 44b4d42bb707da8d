#!/usr/bin/env python3
"""The timing gate of the prefill launch plan as a TOOL (its output belongs in profiles/): runs tests/test_perf_plan_gate.py
(`pytest -m perf`) and prints one line per shape.  usage: python tools/plan_gate.py > profiles/rNN_plan_gate.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_perf_plan_gate.py"), "-m", "perf", "-q", "-s", "-p", "no:cacheprovider"], cwd=ROOT))
