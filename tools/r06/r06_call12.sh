#!/bin/bash
# round 6, GPU call 12: single TP8 prompts — the planner against every forced piece length; the N = 2 line dry; the N = 1 scale_series leg
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c12; mkdir -p $O
timeout 900 python tools/lab/tp8_single_prompt_sweep.py > $O/tp8_single_prompt_sweep.txt 2>&1; echo "rc=$?" >> $O/tp8_single_prompt_sweep.txt; grep -v amdgpu.ids $O/tp8_single_prompt_sweep.txt
for n in 2 4 8; do
  VATTN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > $O/bench_n${n}_gloo_dry.json 2> $O/bench_n${n}_gloo_dry.err
  echo "gloo$n rc=$? lines=$(wc -l < $O/bench_n${n}_gloo_dry.json)"; python -c "
import json,sys; d=json.loads(open('$O/bench_n${n}_gloo_dry.json').read()); print({k:d.get(k) for k in ('value','n_gpus','scaling','tensor_parallel')}); print(d.get('scaling_reference')); print(d.get('legs',{}).get('scale_series'))"
done
timeout 900 python bench.py --leg scale_series > $O/scale_series_n1.json 2> $O/scale_series_n1.err; echo "leg rc=$?"; cat $O/scale_series_n1.json | head -c 900
