#!/bin/bash
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c9; mkdir -p $O
bash tools/r05/r05_planner_ab.sh
timeout 420 python tools/ref_wrapper_bench.py dynamic_tp8 > $O/ref_wrapper_bench.txt 2>&1; tail -8 $O/ref_wrapper_bench.txt | cut -c1-400
