#!/bin/bash
# GPU call D of round 2: prefill64 with fma+exp softmax (no pre-scale, no accumulator pre-fill), fused RoPE tests, timing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_rope_fusion.py -m gpu -q --timeout 300 > gpurun_out/d2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/d2_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/d2_tests.log | tail -20
V=14
timeout 300 python tools/kbench.py prefill --variants 0,$V > gpurun_out/d3_kbench.log 2>&1
cat gpurun_out/d3_kbench.log
timeout 300 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k" \
    --variants $((V + 1024)),$((V + 1280)),$((V + 1536)),$((V + 1792)),$((V + 2048)),$((V + 2304)) > gpurun_out/d4_kbench_ablations.log 2>&1
cat gpurun_out/d4_kbench_ablations.log
