#!/bin/bash
# GPU call I of round 2: single-launch merges (decode, KV-split prefill), two head blocks per decode workgroup, the tensor-parallel
# workloads of bench.py (one rank's share on this GPU).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --timeout 600 > gpurun_out/i1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/i1_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=\|SKIPPED" gpurun_out/i1_tests.log | tail -20
for v in 0 256 512 128; do
  echo "#### decode variant $v (0 default, 256 merge by a second launch, 512 merge always inside the launch, 128 one head block per workgroup)"
  timeout 200 python tools/kbench.py decode --variant $v 2>&1 | grep -v amdgpu
done > gpurun_out/i2_kbench_decode.log 2>&1
cat gpurun_out/i2_kbench_decode.log
for v in 0 16384; do
  echo "#### prefill variant $v (0 default: key-range shares merged inside the launch, 16384: combine_rows_kernel)"
  timeout 200 python tools/kbench.py prefill --only "tp8 8k,tp8 4k,tp8 2k,chunk2k@30k,chunk512@16k,chunk1k@64k,chunk512@8k" --variants $v 2>&1 | grep -v amdgpu
done > gpurun_out/i3_kbench_prefill_merge.log 2>&1
cat gpurun_out/i3_kbench_prefill_merge.log
timeout 600 python bench.py --rank-of 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/i4_bench_rank_of_2.json 2> gpurun_out/i4.err
tail -c 1500 gpurun_out/i4_bench_rank_of_2.json; tail -3 gpurun_out/i4.err
timeout 600 python bench.py --rank-of 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/i5_bench_rank_of_8.json 2> gpurun_out/i5.err
tail -c 1500 gpurun_out/i5_bench_rank_of_8.json; tail -3 gpurun_out/i5.err
