#!/bin/bash
# A/B of the prefill kernels on ONE box (clocks and boxes differ by a few percent, so both libraries are timed back to back):
#   bash tools/p64_ab.sh [tests|notests] — parity tests with the working-tree library, then tools/kbench.py prefill with it (A), with
#   build/base/libvattn_amd.so (B: the library of a commit, tools/build_base.py), and A again.
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O
ONLY="yi6b whole,yi6b chunk4k@28k,llama8b 16k,llama70b/tp8 8k,llama70b/tp8 chunk2k@30k"
if [ "${1:-tests}" = "tests" ]; then
    timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_parity.py -m gpu -q --timeout 300 \
        -k "prefill_chunk_parity or kv_split or variable_length or rescale or work_list or prefill64_midsize or virtual or 4_gib or sampled_blocks or megacache_views" > $O/tests.log 2>&1
    echo "tests rc=$?" >> $O/tests.log
    tail -4 $O/tests.log
fi
run() {
    timeout 300 python tools/kbench.py prefill --variant 0 --only "$ONLY" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
    timeout 300 python tools/kbench.py prefill --variant 14 --worklist --only "llama70b/tp8 8k,llama70b/tp8 4k,chunk512@16k,chunk2k@30k,llama70b/tp8 2k" 2>&1 | grep -v "^--\|^==\|amdgpu.ids"
}
echo "A (working tree)"; run | tee $O/a1.txt
cp vattention_amd/libvattn_amd.so /tmp/new.so
if [ -f build/base/libvattn_amd.so ]; then
    cp build/base/libvattn_amd.so vattention_amd/libvattn_amd.so
    echo "B (last commit)"; run | tee $O/b.txt
    cp /tmp/new.so vattention_amd/libvattn_amd.so
    echo "A again"; run | tee $O/a2.txt
fi
