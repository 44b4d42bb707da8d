#!/bin/bash
# round 6, GPU call 14: the Sarathi hybrid leg with chunks that leave CUs free (1 024 / 512 tokens): serial order vs the two-stream policy
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c14; mkdir -p $O
for leg in hybrid_sarathi_1k_chunks hybrid_sarathi_512_chunks hybrid_sarathi; do
  timeout 600 python bench.py --leg $leg > $O/$leg.json 2> $O/$leg.err; echo "$leg rc=$?"; cat $O/$leg.json
done
