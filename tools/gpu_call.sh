#!/bin/bash
# One GPU call (gpurun -- bash tools/gpu_call.sh STEP...): runs the named steps in order, every step under its own timeout, logs under
# gpurun_out/<tag>_*.  Steps:
#   tests[:EXPR]      pytest -m gpu (optionally -k EXPR)
#   smoke             __graft_entry__.smoke()
#   bench[:ARGS]      python bench.py ARGS            (default: --steps 2 --warmup 1)
#   rank:N[:ARGS]     python bench.py --rank-of N ARGS (default: --steps 1 --warmup 1)
#   kbench[:ARGS]     python tools/kbench.py ARGS
#   py:FILE[:ARGS]    python FILE ARGS
#   sh:FILE[:ARGS]    bash FILE ARGS
#   gloo2[:ARGS]      python bench.py --gpus 2 ARGS with VATTN_BENCH_BACKEND=gloo (two ranks sharing the one GPU: the N > 1 code path)
#   prof:NAME:CMD     rocprofv3 --kernel-trace --stats of CMD, summary copied to gpurun_out/<tag>_prof_NAME/
# TAG (environment) prefixes the log names (default "c").
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${TAG:-c}
i=0
for step in "$@"; do
    i=$((i + 1))
    kind=${step%%:*}
    rest=${step#*:}
    [ "$rest" = "$step" ] && rest=""
    log=gpurun_out/${TAG}${i}_${kind}.log
    t0=$(date +%s)
    case $kind in
        tests)
            if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -q -rs --timeout 900 -x -k "$rest" > $log 2>&1
            else timeout 1800 python -m pytest tests -m gpu -q -rs --timeout 900 > $log 2>&1; fi
            echo "tests rc=$?" >> $log
            grep -n "AssertionError:\|Error\|passed\|failed\|rc=\|SKIPPED" $log | tail -12 ;;
        smoke)
            timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $log 2>&1; echo "smoke rc=$?" >> $log; tail -2 $log ;;
        bench)
            timeout 1500 python bench.py ${rest:---steps 2 --warmup 1} > gpurun_out/${TAG}${i}_bench.json 2> $log
            echo "bench rc=$?" >> $log; tail -3 $log; tail -c 1500 gpurun_out/${TAG}${i}_bench.json ;;
        rank)
            n=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args="--steps 1 --warmup 1"
            timeout 900 python bench.py --rank-of $n $args > gpurun_out/${TAG}${i}_rank${n}.json 2> $log
            echo "rank rc=$?" >> $log; tail -3 $log; tail -c 1200 gpurun_out/${TAG}${i}_rank${n}.json ;;
        kbench)
            timeout 900 python tools/kbench.py $rest > $log 2>&1; echo "kbench rc=$?" >> $log; cat $log | tail -60 ;;
        py)
            f=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
            timeout 900 python $f $args > $log 2>&1; echo "py rc=$?" >> $log; tail -40 $log ;;
        sh)
            f=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
            timeout 1200 bash $f $args > $log 2>&1; echo "sh rc=$?" >> $log; tail -60 $log ;;
        gloo2)
            VATTN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 ${rest:---steps 1 --warmup 1} > gpurun_out/${TAG}${i}_gloo2.json 2> $log
            echo "gloo2 rc=$?" >> $log; tail -3 $log; tail -c 1500 gpurun_out/${TAG}${i}_gloo2.json ;;
        prof)
            name=${rest%%:*}; cmd=${rest#*:}
            ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- bash -c "cd $GRAFT_REPO_ROOT && $cmd" ) > $log 2>&1
            echo "prof rc=$?" >> $log
            mkdir -p gpurun_out/${TAG}${i}_prof_$name
            find /tmp/prof_$name -name "*stats*" -exec cp {} gpurun_out/${TAG}${i}_prof_$name/ \; 2>/dev/null
            ls gpurun_out/${TAG}${i}_prof_$name | head ;;
        *) echo "unknown step $step" ;;
    esac
    echo "[step $i $kind: $(( $(date +%s) - t0 )) s]"
done
