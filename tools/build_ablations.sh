#!/bin/bash
# Debug builds of libvattn_amd.so with one prefill phase ablated each (tools/kbench.py --lib <so>); never shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abl
for a in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-value -DVATTN_ABLATE=$a \
    vattention_amd/csrc/page_manager.cpp vattention_amd/csrc/hip_backend.cpp vattention_amd/csrc/capi.cpp vattention_amd/csrc/attn_api.hip vattention_amd/csrc/prefill_kernels.hip vattention_amd/csrc/decode_kernels.hip vattention_amd/csrc/cache_kernels.hip \
    -o tools/abl_$a.so &
done
wait
ls -la tools/abl_*.so
