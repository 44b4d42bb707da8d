#!/bin/bash
# GPU call J of round 2: fence-free in-launch merges (device-scope stores / loads), capped TP8 workload.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_hybrid_fused.py tests/test_gpu_fuzz.py -m gpu -q --timeout 300 -k "merge or kv_split or hybrid or fuzz or decode" > gpurun_out/j1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/j1_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/j1_tests.log | tail -12
for v in 0 1024 512; do
  echo "#### decode variant $v (0 default = merge by a second launch, 1024 in-launch merge with device-scope accesses, 512 in-launch merge with fences)"
  timeout 200 python tools/kbench.py decode --variant $v 2>&1 | grep -v amdgpu
done > gpurun_out/j2_kbench_decode.log 2>&1
cat gpurun_out/j2_kbench_decode.log
for v in 0 32768; do
  echo "#### prefill variant $v (0 default: combine_rows_kernel, 32768: key-range shares merged inside the launch, device-scope accesses)"
  timeout 200 python tools/kbench.py prefill --only "tp8 8k,tp8 4k,tp8 2k,chunk2k@30k,chunk512@16k,chunk1k@64k,chunk512@8k" --variants $v 2>&1 | grep -v amdgpu
done > gpurun_out/j3_kbench_prefill_merge.log 2>&1
cat gpurun_out/j3_kbench_prefill_merge.log
timeout 300 python tools/hybrid_probe.py > gpurun_out/j4_hybrid_probe.txt 2>&1
grep -v amdgpu gpurun_out/j4_hybrid_probe.txt
timeout 600 python bench.py --rank-of 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/j5_bench_rank_of_8.json 2> gpurun_out/j5.err
tail -c 900 gpurun_out/j5_bench_rank_of_8.json; tail -3 gpurun_out/j5.err
