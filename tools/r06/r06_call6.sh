#!/bin/bash
# round 6, GPU call 6: row sums on the matrix pipe (16 v_mfma_f32_4x4x4 per tile instead of 64 v_add_f32)
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c6; mkdir -p $O
KBENCH_LAB_SUB=12 timeout 600 python tools/p64_variants.py r6:1,r6:5,r6:7 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; grep -v amdgpu.ids $O/p64_variants.txt
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_msum.txt "1 5" "12"
