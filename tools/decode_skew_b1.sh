#!/bin/bash
# per-kv-head finish times at chip-filling sizes: is the head whose bytes have address bits [9:8] = 01 the last one there too?
cd "$(dirname "$0")/.."; export PYTHONUNBUFFERED=1
for a in "16 32768 32 4" "1 131072 28 4 -192" "4 131072 28 4" "64 9500 8 1" "16 32768 32 8"; do
  timeout 200 python tools/decode_skew_probe.py $a 2>&1 | grep "^shape\|exit by kv head (mean, max)\|exit MEAN\|^==" | cut -c1-200
done
