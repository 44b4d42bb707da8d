#!/bin/bash
# round 6, GPU call 22 (final code): the kernel microbenchmark tables — every shape on the default plan (prefill TFLOP/s, decode GB/s), the main shapes in bf16
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c22; mkdir -p $O
timeout 900 python tools/kbench.py --variants 0 2>&1 | grep -v amdgpu.ids > $O/kbench.txt; tail -60 $O/kbench.txt
timeout 600 python tools/kbench.py --bf16 --variants 0 --only "yi6b whole,yi6b chunk4k@28k,llama8b 16k,yi34b/tp2 chunk16k@112k,yi6b B16@32k,llama70b/tp8 chunk2k@30k" 2>&1 | grep -v amdgpu.ids > $O/kbench_bf16.txt; cat $O/kbench_bf16.txt
