#!/bin/bash
# GPU call G of round 2: fused prefill||decode launch (parity, probe, e2e), padded-K builds of prefill64, dynamic leg with multi-request look-ahead.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hybrid_fused.py -m gpu -q --timeout 300 > gpurun_out/g1_hybrid_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g1_hybrid_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/g1_hybrid_tests.log | tail -15
timeout 300 python tools/kbench.py prefill --only "yi6b whole,chunk4k@28k,chunk16k@112k,llama8b 16k" --variants 14,526,2574,14,526 > gpurun_out/g2_kbench_kpad.log 2>&1
grep -v amdgpu gpurun_out/g2_kbench_kpad.log
timeout 300 python tools/hybrid_probe.py > gpurun_out/g3_hybrid_probe.txt 2>&1
grep -v amdgpu gpurun_out/g3_hybrid_probe.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_hybrid_fused.py > gpurun_out/g4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g4_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=" gpurun_out/g4_tests.log | tail -12
(timeout 200 python tools/hybrid_e2e.py; timeout 200 python tools/hybrid_e2e.py --ctx 8192 --chunk 512 --batch 64 --pd 10) > gpurun_out/g5_hybrid_e2e.txt 2>&1
grep -v amdgpu gpurun_out/g5_hybrid_e2e.txt
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/g6_bench.log 2> gpurun_out/g6_bench.err
tail -1 gpurun_out/g6_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'roofline',d['roofline']['frac'],'dynamic',d.get('dynamic'))"
