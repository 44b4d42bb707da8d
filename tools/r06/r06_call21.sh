#!/bin/bash
# round 6, GPU call 21 (final code): the profiling recipe once more (rocprofv3 kernel stats of the bench workload + the counter passes -> traffic.json),
# the N = 2 / 8 lines dry over the gloo hook after the bench changes, and the lab suite after the row-sum helper moved
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c21; mkdir -p $O
bash tools/prof_round.sh $O/prof > $O/prof_round.log 2>&1; tail -45 $O/prof_round.log
for n in 2 8; do
  VATTN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > $O/bench_n${n}_gloo_dry.json 2> $O/bench_n${n}_gloo_dry.err
  echo "gloo$n rc=$? lines=$(wc -l < $O/bench_n${n}_gloo_dry.json)"; python -c "
import json,sys; d=json.loads(open('$O/bench_n${n}_gloo_dry.json').read()); print({k:d.get(k) for k in ('value','n_gpus','scaling','tensor_parallel')}); print(d.get('scaling_reference')); print(d.get('legs',{}).get('scale_series')); print(sorted(d['roofline'].get('other',{}).keys()))"
done
timeout 900 python -m pytest tests -m "gpu and lab" -q --timeout 600 > $O/tests_lab.log 2>&1; echo "lab rc=$?" >> $O/tests_lab.log; tail -4 $O/tests_lab.log
