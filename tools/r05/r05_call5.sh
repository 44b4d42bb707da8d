#!/bin/bash
# Round 5, GPU call 5: persistent workgroups with DRAWN queues — parity, kbench A/B, then the two dynamic legs persistent vs per piece, alternating
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_prefill_persistent.py -m gpu -q -x --timeout 300 > $O/tests_persistent.log 2>&1; echo "persistent tests rc=$?" | tee -a $O/tests_persistent.log; tail -4 $O/tests_persistent.log | cut -c1-300
grep -q "rc=0" $O/tests_persistent.log || exit 0
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m gpu -q --timeout 600 -k "work_list or fuzz" > $O/tests_more.log 2>&1; echo "more tests rc=$?" | tee -a $O/tests_more.log; tail -4 $O/tests_more.log | cut -c1-300
SH="llama70b/tp8 8k,llama70b/tp8 4k,chunk2k@30k,llama8b 16k,small 2k,chunk1k@64k,llama8b chunk512@8k"
for i in 1 2; do
  echo "== A persistent =="; timeout 300 python tools/kbench.py prefill --variant 0 --worklist --only "$SH" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
  echo "== B per piece ==";  timeout 300 python tools/kbench.py prefill --variant 0 --worklist --per-piece --only "$SH" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
done | tee $O/kbench_ab.txt
for i in 1 2; do
  for leg in dynamic_tp8_rank dynamic; do
    for mode in "" "--per-piece-prefill"; do
      tag=${leg}_${i}_$( [ -z "$mode" ] && echo persistent || echo per_piece )
      timeout 600 python bench.py --leg $leg $mode > $O/$tag.json 2> $O/$tag.err
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); e = list(d.values())[0]
w = e["warm_pool_pass"]
print("%-40s tokens/s fresh %9.1f warm %9.1f | prefill frac fresh %.4f (%.4f ms x %d) warm %.4f (%.4f ms) | decode warm %.4f" % (
    sys.argv[2], e["tokens_per_s"], w["tokens_per_s"], e["roofline_prefill"]["frac"], e["roofline_prefill"]["ms_per_launch"], e["roofline_prefill"]["launches"],
    w["roofline_prefill"]["frac"], w["roofline_prefill"]["ms_per_launch"], w["roofline_decode"]["frac"]))
PY
    done
  done
done | tee $O/legs_ab.txt
