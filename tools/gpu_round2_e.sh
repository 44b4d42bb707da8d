#!/bin/bash
# GPU call E of round 2: schedule variants of prefill64 (exp split, fragment ring depth), fused RoPE tests, bench with the dynamic leg fixed.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_rope_fusion.py tests/test_gpu_attention.py -m gpu -q --timeout 300 -k "rope or rotary or dma or deferred" > gpurun_out/e2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/e2_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/e2_tests.log | tail -12
V=14
timeout 300 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k,chunk16k@112k,tp8 8k,small 2k" \
    --variants 0,$V,$((V + 256)),$((V + 512)),$((V + 768)),$((V + 2816)) > gpurun_out/e3_kbench_variants.log 2>&1
grep -v amdgpu gpurun_out/e3_kbench_variants.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/e5_bench.log 2>&1
tail -1 gpurun_out/e5_bench.log
