#!/bin/bash
# round 6, GPU call 18: the power ceiling of whole-chip MFMA streams (tools/lab/power_ceiling.py) and prefill32 now inside the lab library
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/c18
mkdir -p $O
timeout 300 python tools/lab/power_ceiling.py 2>&1 | grep -v amdgpu.ids | tee $O/power_ceiling.txt
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m "gpu" -q --timeout 300 -x \
    -k "prefill_chunk_parity or kv_split or variable_length or rescale or workgroup_orders or fuzz_prefill" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for v in 14 6; do timeout 300 python tools/kbench.py prefill --variant $v --only "yi6b whole,llama8b 16k" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"; done | tee $O/ab.txt
