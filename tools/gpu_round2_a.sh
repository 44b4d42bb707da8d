#!/bin/bash
# GPU call A of round 2: the whole GPU suite without the new kernel, hardware self-tests, then the new prefill64 kernel
# (separate processes: a GPU fault must not take the other results with it), then kernel timings.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T="timeout 900"
$T python -m pytest tests -m gpu -q --timeout 600 -k "not dma and not deferred_rescale and not fuzz" > gpurun_out/a1_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/a1_suite.log
tail -5 gpurun_out/a1_suite.log
timeout 120 python - > gpurun_out/a2_selftest.log 2>&1 <<'PY'
import torch
from vattention_amd import kernels
rc, d = kernels.selftest_layouts(torch.device("cuda:0"))
print("selftest rc", rc, "detail", d, kernels.last_error() if rc else "")
PY
cat gpurun_out/a2_selftest.log
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q --timeout 120 -k "dma or deferred_rescale" > gpurun_out/a3_new_kernel.log 2>&1
echo "new kernel rc=$?" >> gpurun_out/a3_new_kernel.log
tail -30 gpurun_out/a3_new_kernel.log
timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 200 > gpurun_out/a4_fuzz.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/a4_fuzz.log
tail -5 gpurun_out/a4_fuzz.log
timeout 300 python tools/kbench.py prefill --variants 0,14 > gpurun_out/a5_kbench.log 2>&1
cat gpurun_out/a5_kbench.log
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/a6_bench.log 2>&1
tail -3 gpurun_out/a6_bench.log
