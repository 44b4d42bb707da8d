#!/bin/bash
# GPU call H of round 2: KV-split sweep for underfilled causal prompts (TP8 shards), fused launch with one fence per workgroup, early P packing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 200 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k,chunk16k@112k,llama8b 16k" --variants 0,782,0,782 > gpurun_out/h1_kbench_earlypack.log 2>&1
grep -v amdgpu gpurun_out/h1_kbench_earlypack.log
for sp in 0 1 2 3 4; do
  echo "#### forced prefill KV splits: $sp (0 = plan)"
  timeout 120 python tools/kbench.py prefill --only "tp8 8k,tp8 4k,tp8 2k,small 2k,chunk4k@0" --variants 14,8,2 --pf-splits $sp 2>&1 | grep -v amdgpu
done > gpurun_out/h2_kbench_split_sweep.log 2>&1
cat gpurun_out/h2_kbench_split_sweep.log
timeout 300 python tools/hybrid_probe.py > gpurun_out/h3_hybrid_probe.txt 2>&1
grep -v amdgpu gpurun_out/h3_hybrid_probe.txt
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_hybrid_fused.py tests/test_gpu_rope_fusion.py -m gpu -q --timeout 300 > gpurun_out/h4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/h4_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/h4_tests.log | tail -12
