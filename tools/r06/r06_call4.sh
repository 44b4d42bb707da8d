#!/bin/bash
# round 6, GPU call 4: does the ENCODING SIZE price a filler?  v_fma -> v_mul in VOP2 (4 B) / VOP3 (8 B) (wrong results), and the exact
# 4-byte form v_mov + v_fmac; v_max3 -> two v_max
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c4; mkdir -p $O
timeout 600 python tools/p64_variants.py r6:1,r6:5,r6:2,r6:4,r6:6,r6:7 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; grep -v amdgpu.ids $O/p64_variants.txt
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_variants.txt "1 5 2 4 6 7"
