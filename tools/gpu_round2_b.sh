#!/bin/bash
# GPU call B of round 2: prefill64 fixes (combine launch, exact scale), its parity tests, timing decomposition by ablation builds.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 120 python - > gpurun_out/b1_selftest.log 2>&1 <<'PY'
import torch
from vattention_amd import kernels
rc, d = kernels.selftest_layouts(torch.device("cuda:0"))
print("selftest rc", rc, "detail", d, kernels.last_error() if rc else "")
PY
cat gpurun_out/b1_selftest.log
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_parity.py -m gpu -q --timeout 300 \
    -k "dma or deferred_rescale or orders or kv_split or batched or fuzz or decode_b16 or decode_b8" > gpurun_out/b2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/b2_tests.log
grep -n "AssertionError:\|passed\|failed\|rc=" gpurun_out/b2_tests.log | tail -20
V=14
timeout 300 python tools/kbench.py prefill --variants $V,$((V + 256)) > gpurun_out/b3_kbench_exact_vs_prescale.log 2>&1
cat gpurun_out/b3_kbench_exact_vs_prescale.log
timeout 300 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k,small 2k" \
    --variants 0,$((V + 512)),$((V + 768)),$((V + 2560)),$((V + 1024)),$((V + 1280)),$((V + 1536)),$((V + 1792)),$((V + 2048)),$((V + 2304)) > gpurun_out/b4_kbench_ablations.log 2>&1
cat gpurun_out/b4_kbench_ablations.log
