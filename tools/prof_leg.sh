#!/bin/bash
# rocprofv3 kernel statistics of one bench leg (bash tools/prof_leg.sh LEG OUTDIR): per-kernel share of the leg's GPU time
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LEG=${1:-dynamic_tp8_rank}; O=${2:-gpurun_out/prof_leg}
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --leg $LEG > $O/${LEG}.json 2> $O/${LEG}.err
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/${LEG}_kernel_stats.md 2>> $O/${LEG}.err
rm -rf $O/kt
head -14 $O/${LEG}_kernel_stats.md | cut -c1-220
