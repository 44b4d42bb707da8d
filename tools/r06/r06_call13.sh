#!/bin/bash
# round 6, GPU call 13: the round's profile (rocprofv3 kernel statistics of the bench workload, HBM traffic and SQ counters in separate passes,
# traffic.json with matrix-pipe duty and kernel durations), then the default bench line with every leg
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash tools/prof_round.sh gpurun_out/prof6 > gpurun_out/prof6_tail.txt 2>&1; tail -25 gpurun_out/prof6_tail.txt; cat gpurun_out/prof6/traffic.json
cp gpurun_out/prof6/traffic.json profiles/r06_traffic.json
timeout 1500 python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1_details.log; echo "bench rc=$?"; tail -c 2500 gpurun_out/r06_bench_n1.json
