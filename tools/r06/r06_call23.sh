#!/bin/bash
# round 6, GPU call 23 (HEAD): pytest -m gpu as the driver runs it + smoke(), after the last header clean-ups
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c23; mkdir -p $O
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 900 > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_gpu.log; tail -6 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
