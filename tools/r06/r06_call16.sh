#!/bin/bash
# round 6, GPU call 16: first run of prefill32_kernel (8 waves x 32 rows on prefill64's data flow): parity, then same-box A/B against prefill64
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/c16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m "gpu and not lab" -q --timeout 300 -x \
    -k "prefill_chunk_parity or kv_split or variable_length or rescale or workgroup_orders or fuzz_prefill" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
ONLY="yi6b whole,yi6b chunk4k@28k,llama8b 16k,llama70b/tp8 8k,llama70b/tp8 chunk2k@30k"
for v in 14 6 14 6; do
    echo "== variant $v"
    timeout 300 python tools/kbench.py prefill --variant $v --only "$ONLY" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
done | tee $O/ab.txt
