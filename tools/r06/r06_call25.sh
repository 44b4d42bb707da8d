#!/bin/bash
# round 6, GPU calls 20 and 25 (final code): the driver's suite as one run (pytest -m gpu), smoke, the bench line with the measured ceilings
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c25; mkdir -p $O
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_gpu.log; tail -16 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench_details.log; echo "bench rc=$?"
tail -c 3000 $O/bench.json
