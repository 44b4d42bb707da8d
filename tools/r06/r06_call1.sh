#!/bin/bash
# round 6, GPU call 1: docstring pins on the kernels; prefill64 with the V^T pre-read (bit-identical? faster?); stamps; N = 4 / 8 lines dry over gloo
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c1; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_docstring_pins.py -m gpu -q --timeout 200 > $O/pins.log 2>&1; echo "pins rc=$?" >> $O/pins.log; tail -3 $O/pins.log
timeout 600 python tools/p64_variants.py r6:1 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; cat $O/p64_variants.txt | grep -v amdgpu.ids
timeout 600 python tools/lab/p64_stamps.py > $O/p64_stamps.txt 2>&1; echo "rc=$?" >> $O/p64_stamps.txt; cat $O/p64_stamps.txt | grep -v amdgpu.ids
for n in 4 8; do
  VATTN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > $O/gloo$n.json 2> $O/gloo$n.err
  echo "gloo$n rc=$? lines=$(wc -l < $O/gloo$n.json)"; tail -c 600 $O/gloo$n.json; grep -v "details" $O/gloo$n.err | tail -5
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-dynamic --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?"; tail -c 1200 $O/bench_quick.json
