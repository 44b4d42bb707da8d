#!/bin/bash
# round 6, GPU call 7: DMA piece distances in the scalar offset (6 v_mad fewer per tile), row-max chain without its -inf moves
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c7; mkdir -p $O
KBENCH_LAB_SUB=13 timeout 600 python tools/p64_variants.py r6:1,r6:5,r6:7 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; grep -v amdgpu.ids $O/p64_variants.txt
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_soffset.txt "1 5" "13"
