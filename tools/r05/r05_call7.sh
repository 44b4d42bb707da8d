#!/bin/bash
# Round 5, GPU call 7: the "chunks" policy of the persistent queues inside a replay (one TP2 rank's 128 k request in 16 k chunks), alternating
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c7; mkdir -p $O
for i in 1 2; do
  for mode in "" "--per-piece-prefill" "--persistent-prefill"; do
    tag=c4_${i}_$( [ -z "$mode" ] && echo policy_chunks || echo ${mode#--} )
    timeout 600 python bench.py --leg c4_rank_share_128k $mode > $O/$tag.json 2> $O/$tag.err
    python3 - $O/$tag.json $tag <<'PY'
import json, sys
e = list(json.load(open(sys.argv[1])).values())[0]
print("%-34s tokens/s %9.1f  %.3f s | prefill frac %.4f (%.4f ms x %d) | decode frac %.4f (%.4f ms)" % (
    sys.argv[2], e["tokens_per_s"], e["seconds"], e["prefill_frac"], e["roofline_prefill"]["ms_per_launch"], e["roofline_prefill"]["launches"], e["decode_frac"], e["roofline_decode"]["ms_per_launch"]))
PY
  done
done | tee $O/c4_ab.txt
