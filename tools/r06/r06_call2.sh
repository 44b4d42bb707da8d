#!/bin/bash
# round 6, GPU call 2: stamps kept in LDS (no store in front of the step's vmcnt(0)); issue-budget ablations, timed and counted
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c2; mkdir -p $O
timeout 600 python tools/lab/p64_stamps.py > $O/p64_stamps.txt 2>&1; echo "rc=$?" >> $O/p64_stamps.txt; grep -v amdgpu.ids $O/p64_stamps.txt
timeout 600 python tools/p64_variants.py r6:1,r6:4,r6:5,r6:6,r6:7 > $O/p64_variants.txt 2>&1; echo "rc=$?" >> $O/p64_variants.txt; grep -v amdgpu.ids $O/p64_variants.txt
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_variants.txt
for n in 4 8; do
  VATTN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > $O/gloo$n.json 2> $O/gloo$n.err
  echo "gloo$n rc=$? lines=$(wc -l < $O/gloo$n.json)"; python -c "
import json,sys; d=json.loads(open('$O/gloo$n.json').read()); print({k:d.get(k) for k in ('value','n_gpus','scaling','scaling_reference')}); print(d.get('legs',{}).get('scale_series'))"
done
