#!/bin/bash
# round 6, GPU call 10: the product-only suite with the lab library refused, slowest tests listed
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c10; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m "gpu and not lab" -q --timeout 900 --durations=40 > $O/tests_product.log 2>&1; echo "product rc=$? wall=$(( $(date +%s) - t0 )) s" >> $O/tests_product.log; tail -70 $O/tests_product.log
