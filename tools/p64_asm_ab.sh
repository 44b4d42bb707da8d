#!/bin/bash
# Round 4, prefill64 energy pass (ii): the product library with the compiler's `s_nop 0` between inline-asm statements removed from
# prefill64_kernel's device code (tools/lab/build_asm_ab.sh -> build/asm_ab/) against the same assembly reassembled unchanged, same
# box, alternating; prefill parity tests with the stripped library installed.  The tree's own library is restored at the end.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
ONLY="yi6b whole,yi6b chunk4k@28k,llama8b 16k"
cp vattention_amd/libvattn_amd.so /tmp/product.so
for i in 1 2 3; do
  for v in asis nonop; do
    cp build/asm_ab/libvattn_amd_$v.so vattention_amd/libvattn_amd.so
    echo "== $v (pass $i)"; timeout 200 python tools/kbench.py prefill --variant 0 --only "$ONLY" 2>&1 | grep "TFLOP"
  done
done
cp /tmp/product.so vattention_amd/libvattn_amd.so
echo "== product library as built by build.py"; timeout 200 python tools/kbench.py prefill --variant 0 --only "$ONLY" 2>&1 | grep "TFLOP"
cp build/asm_ab/libvattn_amd_nonop.so vattention_amd/libvattn_amd.so
echo "== parity with the stripped library installed"
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_parity.py -m gpu -q --timeout 300 \
    -k "prefill_chunk_parity or kv_split or variable_length or rescale or work_list or prefill64_midsize or sampled_blocks or megacache_views or operator_by_composition or pod_sweep" 2>&1 | tail -3
cp /tmp/product.so vattention_amd/libvattn_amd.so
