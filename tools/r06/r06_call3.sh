#!/bin/bash
# round 6, GPU call 3: why does the v_fma cost 3.5 cycles where the v_add costs 0.9?  scale in a scalar register / stage order inside a group
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c3; mkdir -p $O
timeout 600 python tools/p64_variants.py r6:1,r6:5,r6:6,r6:7 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; grep -v amdgpu.ids $O/p64_variants.txt
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_variants.txt "1 5 6 7"
