#!/bin/bash
# round 6, GPU call 9: after the laboratory moved out of the product sources — the PRODUCT-ONLY GPU suite (timed; which libraries it maps), then the lab suite
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c9; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m "gpu and not lab" -q --timeout 900 > $O/tests_product.log 2>&1; echo "product rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_product.log; tail -6 $O/tests_product.log
t0=$(date +%s)
timeout 1500 python -m pytest tests -m "gpu and lab" -q --timeout 900 > $O/tests_lab.log 2>&1; echo "lab rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_lab.log; tail -6 $O/tests_lab.log
