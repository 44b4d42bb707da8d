#!/bin/bash
# Round 5, GPU call 4: persistent work lists inside the dynamic legs, same box, alternating (bench.py --leg X [--per-piece-prefill])
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c4; mkdir -p $O
for i in 1 2; do
  for leg in dynamic_tp8_rank dynamic; do
    for mode in "" "--per-piece-prefill"; do
      tag=${leg}_${i}_$( [ -z "$mode" ] && echo persistent || echo per_piece )
      timeout 600 python bench.py --leg $leg $mode > $O/$tag.json 2> $O/$tag.err
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); e = list(d.values())[0]
w = e["warm_pool_pass"]
print("%-40s tokens/s fresh %9.1f warm %9.1f | prefill frac fresh %.4f (%.4f ms x %d) warm %.4f (%.4f ms) | decode warm %.4f" % (
    sys.argv[2], e["tokens_per_s"], w["tokens_per_s"], e["roofline_prefill"]["frac"], e["roofline_prefill"]["ms_per_launch"], e["roofline_prefill"]["launches"],
    w["roofline_prefill"]["frac"], w["roofline_prefill"]["ms_per_launch"], w["roofline_decode"]["frac"]))
PY
    done
  done
done | tee $O/legs_ab.txt
