#!/bin/bash
# the N = 2 code path of bench.py (two ranks sharing the one GPU over the gloo hook): stdout must carry rank 0's ONE JSON line
cd "$(dirname "$0")/.."
VATTN_BENCH_BACKEND=gloo timeout 65 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/n2.out 2> /tmp/n2.err; echo "rc=$?"
echo "stdout lines: $(wc -l < /tmp/n2.out), bytes: $(wc -c < /tmp/n2.out)"; head -c 700 /tmp/n2.out; echo; tail -3 /tmp/n2.err | cut -c1-300
