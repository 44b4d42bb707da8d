#!/bin/bash
# Round 5, GPU call 3: persistent kernel with the pre-based descriptors (no re-base code inside the step) — parity, A/B, then the g1
# table and one short bench run with the new legs.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_prefill_persistent.py -m gpu -q -x --timeout 300 > $O/tests_persistent.log 2>&1; echo "persistent tests rc=$?" | tee -a $O/tests_persistent.log; tail -4 $O/tests_persistent.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 -k "work_list or fuzz" > $O/tests_more.log 2>&1; echo "more tests rc=$?" | tee -a $O/tests_more.log; tail -4 $O/tests_more.log | cut -c1-300
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
echo "== g1 table =="
timeout 900 bash tools/g1_table.sh > $O/g1_small_pages.md 2>&1; tail -30 $O/g1_small_pages.md | cut -c1-400
echo "== bench, short =="
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
