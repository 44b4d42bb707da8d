#!/bin/bash
# Round 4, prefill64 energy pass (i): non-temporal Q loads / O stores (lab build 15) against the product build inside the lab library
# (build 14), same box, alternating; then the L2 / fabric counters of both (separate --pmc passes).
cd "$(dirname "$0")/.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
A=$(( (14 << 8) | 14 )); B=$(( (15 << 8) | 14 ))
ONLY="yi6b whole,yi6b chunk4k@28k,llama8b 16k"
python tools/p64_variants.py 14,15 2>&1 | tail -4
for i in 1 2 3; do
  echo "== product schedule (lab build 14)"; python tools/kbench.py prefill --variant $A --only "$ONLY" 2>&1 | grep "TFLOP"
  echo "== nt Q loads + O stores (lab build 15)"; python tools/kbench.py prefill --variant $B --only "$ONLY" 2>&1 | grep "TFLOP"
done
for V in $A $B; do
  for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    rm -rf gpurun_out/pmcq; timeout 200 rocprofv3 --pmc $P -d gpurun_out/pmcq -- python tools/kbench.py prefill --variant $V --only "yi6b whole" > /dev/null 2> gpurun_out/pmcq.err
    echo "-- variant $V: $P"; python tools/pmc_summary.py gpurun_out/pmcq prefill64 | grep -v "^_ZN"
  done
done
rm -rf gpurun_out/pmcq
