#!/bin/bash
# Round 5, GPU call 2: the persistent work-list kernel (prefill64p_kernel) — parity first, then same-box A/B against the one-workgroup-
# per-piece launch of the same planner family (tools/kbench.py --worklist [--per-piece]).
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_prefill_persistent.py -m gpu -q -x --timeout 300 > $O/tests_persistent.log 2>&1; echo "persistent tests rc=$?" | tee -a $O/tests_persistent.log; tail -15 $O/tests_persistent.log
if grep -q "rc=0" $O/tests_persistent.log; then
  timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_parity.py -m gpu -q --timeout 600 \
      -k "work_list or fuzz or variable_length or sampled_blocks or tp8 or chunk" > $O/tests_more.log 2>&1; echo "more tests rc=$?" | tee -a $O/tests_more.log; tail -6 $O/tests_more.log
fi
SH="llama70b/tp8 8k,llama70b/tp8 4k,llama70b/tp8 2k,chunk2k@30k,chunk512@16k,llama8b 16k,small 2k,yi6b chunk4k@0,chunk1k@64k,llama8b chunk512@8k"
for i in 1 2; do
  echo "== A persistent =="; timeout 300 python tools/kbench.py prefill --variant 0 --worklist --only "$SH" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
  echo "== B per piece ==";  timeout 300 python tools/kbench.py prefill --variant 0 --worklist --per-piece --only "$SH" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
done | tee $O/kbench_ab.txt
echo "== big shapes: lists forced on (persistent max blocks 1e6) vs default grid =="
for i in 1 2; do
  KBENCH_PERSIST_MAX_BLOCKS=1000000 timeout 300 python tools/kbench.py prefill --variant 0 --worklist --only "yi6b whole,yi6b chunk4k@28k,chunk16k@112k" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
  timeout 300 python tools/kbench.py prefill --variant 0 --only "yi6b whole,yi6b chunk4k@28k,chunk16k@112k" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
done | tee $O/kbench_big.txt
