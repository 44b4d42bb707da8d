#!/bin/bash
# Same-box A/B of the decode path (bash tools/dec_ab.sh [tests]): working-tree library (A), build/base/libvattn_amd.so (B, tools/build_base.py), A again.
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
if [ "${1:-}" = "tests" ]; then
    timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py -m gpu -q --timeout 300 -k "decode" 2>&1 | tail -3
fi
run() { timeout 200 python tools/kbench.py decode --only "yi6b B16@32k,yi6b B1@32k,yi6b B4@32k,llama70b/tp8 B64@32k,llama8b B64@8k" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"; }
echo "A (working tree)"; run
cp vattention_amd/libvattn_amd.so /tmp/new.so
cp build/base/libvattn_amd.so vattention_amd/libvattn_amd.so
echo "B (base)"; run
cp /tmp/new.so vattention_amd/libvattn_amd.so
echo "A again"; run
