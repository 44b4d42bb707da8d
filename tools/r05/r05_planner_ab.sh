#!/bin/bash
# Round 5: the finer cut candidates of the prefill work-list planner (working tree, A) against the planner of the last commit (build/base, B):
# one-prompt launches on a TP8 rank (8 query / 1 kv head) and the replay's admission batches, one workgroup per piece, alternating.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05plan; mkdir -p $O
cp vattention_amd/libvattn_amd.so /tmp/new.so
run() { timeout 300 python tools/p64p_plan_probe.py 2>&1 | grep -v amdgpu.ids; }
for i in 1 2; do
  cp /tmp/new.so vattention_amd/libvattn_amd.so; echo "== A (working tree)"; run
  cp build/base/libvattn_amd.so vattention_amd/libvattn_amd.so; echo "== B (last commit)"; run
done | tee $O/planner_ab.txt
cp /tmp/new.so vattention_amd/libvattn_amd.so
