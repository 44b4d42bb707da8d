#!/bin/bash
# round 6, GPU call 17: prefill32_kernel against prefill64_kernel on one box, and its schedule placements (tools/lab/p32_builds.sh libraries)
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/c17
mkdir -p $O
ONLY="yi6b whole,yi6b chunk4k@28k,llama8b 16k,llama70b/tp8 8k,llama70b/tp8 chunk2k@30k"
kb() { timeout 300 python tools/kbench.py prefill --variant $1 --only "$ONLY" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"; }
cp vattention_amd/libvattn_amd.so /tmp/product.so
{
echo "== prefill64 (variant 14)"; kb 14
echo "== prefill32 (variant 6), product schedule"; kb 6
for f in build/p32/libvattn_*.so; do
    cp $f vattention_amd/libvattn_amd.so
    echo "== prefill32 $(basename $f)"; kb 6
done
cp /tmp/product.so vattention_amd/libvattn_amd.so
echo "== prefill64 (variant 14) again"; kb 14
echo "== prefill32 (variant 6) again"; kb 6
} | tee $O/ab.txt
