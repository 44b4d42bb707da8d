#!/bin/bash
# Round 5, final GPU call: the whole -m gpu suite, the bench line, the round's rocprofv3 evidence (kernel stats of the bench workload and of
# both dynamic legs, PMC traffic and SQ counters), the N = 2 stdout check over the gloo hook, the plan gate, the reference-wrapper bench.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05final; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -rs --timeout 900 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -5 $O/tests.log | cut -c1-300
echo "[tests: $(( $(date +%s) - t0 )) s]"; t0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1200 $O/bench.json
echo "[bench: $(( $(date +%s) - t0 )) s]"; t0=$(date +%s)
timeout 900 bash tools/prof_round.sh $O/prof > $O/prof_round.log 2>&1; echo "prof_round rc=$?"; head -8 $O/prof/bench_kernel_stats.md | cut -c1-200
echo "[prof_round: $(( $(date +%s) - t0 )) s]"; t0=$(date +%s)
timeout 400 bash tools/prof_leg.sh dynamic $O/prof_dynamic > $O/prof_dynamic.log 2>&1; timeout 400 bash tools/prof_leg.sh dynamic_tp8_rank $O/prof_tp8 > $O/prof_tp8.log 2>&1
head -6 $O/prof_dynamic/dynamic_kernel_stats.md | cut -c1-160; head -6 $O/prof_tp8/dynamic_tp8_rank_kernel_stats.md | cut -c1-160
echo "[prof legs: $(( $(date +%s) - t0 )) s]"; t0=$(date +%s)
bash tools/bench_n2_stdout_check.sh > $O/n2_check.log 2>&1; head -3 $O/n2_check.log | cut -c1-400
timeout 300 python tools/plan_gate.py > $O/plan_gate.txt 2>&1; tail -5 $O/plan_gate.txt | cut -c1-200
timeout 600 python tools/ref_wrapper_bench.py dynamic_tp8 > $O/ref_wrapper_bench.txt 2>&1; tail -12 $O/ref_wrapper_bench.txt | cut -c1-250
echo "[rest: $(( $(date +%s) - t0 )) s]"
