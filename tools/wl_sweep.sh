#!/bin/bash
# schedule builds of prefill64 inside the lab library, then a piece-length sweep of the prefill work list on the tensor-parallel shapes
# (bash tools/wl_sweep.sh on a GPU box; results: profiles/r03_p64_schedules.txt, run E)
cd "$(dirname "$0")/.."
python tools/p64_variants.py 14,1,2,12,14,1,2 2>&1 | grep "^yi6b\|^llama"
for T in 0 12 16 22 32 44; do echo "== forced piece length $T (0 = the planner's choice)"; timeout 200 python tools/kbench.py prefill --variant 14 --worklist --wl-tiles $T --only "llama70b/tp8 8k,llama70b/tp8 4k,chunk512@16k,llama70b/tp8 2k" 2>&1 | grep "work list\|tp8 2k"; done
