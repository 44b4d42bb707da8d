#!/bin/bash
# GPU call C of round 2: restructured prefill64 schedule (staged softmax pipeline, mid-phase barrier, DMA in bare groups), parity,
# timing decomposition, and the new bench.py end to end.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m gpu -q --timeout 300 \
    -k "dma or deferred_rescale or orders or kv_split or batched or fuzz" > gpurun_out/c2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c2_tests.log
grep -n "AssertionError:\|passed\|failed\|rc=" gpurun_out/c2_tests.log | tail -20
V=14
timeout 300 python tools/kbench.py prefill --variants $V,$((V + 256)) > gpurun_out/c3_kbench_exact_vs_prescale.log 2>&1
cat gpurun_out/c3_kbench_exact_vs_prescale.log
timeout 300 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k,small 2k" \
    --variants 0,$((V + 1024)),$((V + 1280)),$((V + 1536)),$((V + 1792)),$((V + 2048)),$((V + 2304)) > gpurun_out/c4_kbench_ablations.log 2>&1
cat gpurun_out/c4_kbench_ablations.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/c5_bench.log 2>&1
tail -2 gpurun_out/c5_bench.log
VATTN_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 1 --warmup 1 --layers 2 --ctx 16384 > gpurun_out/c6_bench_tp2_gloo.log 2>&1
tail -2 gpurun_out/c6_bench_tp2_gloo.log
