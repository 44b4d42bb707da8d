#!/bin/bash
# Round 5, last GPU call: the whole -m gpu suite and the bench line on the FINAL code (planner with the finer cut candidates)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05last; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rs --timeout 900 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
