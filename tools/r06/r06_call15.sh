#!/bin/bash
# round 6, GPU call 15: product-only suite (timed, lab library refused), lab suite, smoke, the reference's unmodified wrapper on the TP8 dynamic trace
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c15; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m "gpu and not lab" -q --timeout 900 --durations=12 > $O/tests_product.log 2>&1; echo "product rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_product.log; tail -22 $O/tests_product.log
t0=$(date +%s)
timeout 1500 python -m pytest tests -m "gpu and lab" -q --timeout 900 > $O/tests_lab.log 2>&1; echo "lab rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_lab.log; tail -5 $O/tests_lab.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python tools/ref_wrapper_bench.py dynamic_tp8 > $O/ref_wrapper_bench.txt 2>&1; echo "rc=$?" >> $O/ref_wrapper_bench.txt; grep -v amdgpu.ids $O/ref_wrapper_bench.txt | tail -12
