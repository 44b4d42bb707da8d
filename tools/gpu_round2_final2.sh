#!/bin/bash
# Round-2 closing call: full GPU suite and the bench line on the final code (the manager's look-ahead reclaim and the last-round plan rule
# came after tools/gpu_round2_final.sh), plus the TP4 share that the plan rule targets.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --timeout 600 > gpurun_out/n1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/n1_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=\|SKIPPED" gpurun_out/n1_tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/n3_bench.json 2> gpurun_out/n3_bench.err
tail -c 700 gpurun_out/n3_bench.json; tail -2 gpurun_out/n3_bench.err
timeout 400 python bench.py --rank-of 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/n4_bench_rank_of_4.json 2> gpurun_out/n4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/n4_bench_rank_of_4.json").read().strip().split("\n")[-1])
print("rank-of-4:", d["value"], d["ms_per_step"], d["roofline"])
d = json.loads(open("gpurun_out/n3_bench.json").read().strip().split("\n")[-1])
print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_decode"]["frac"]); print(d["cold_wave"]); print(d["dynamic"])
PY
