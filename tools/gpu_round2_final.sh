#!/bin/bash
# Round-2 final measurement call: full GPU suite, smoke(), the bench line with driver-like flags, rocprofv3 kernel stats of the same
# workload, SQ and HBM counters of the dominant kernels (separate --pmc passes), kernel microbenchmarks, one rank's share of the
# tensor-parallel workloads.  Outputs under gpurun_out/k*; the summaries are copied to profiles/r02_*.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/k
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs --timeout 600 > gpurun_out/k1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/k1_tests.log
grep -n "AssertionError:\|Error\|passed\|failed\|rc=\|SKIPPED" gpurun_out/k1_tests.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/k2_smoke.log 2>&1
tail -2 gpurun_out/k2_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/k3_bench.json 2> gpurun_out/k3_bench.err
tail -c 600 gpurun_out/k3_bench.json; tail -2 gpurun_out/k3_bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/k/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dynamic > gpurun_out/k/bench_prof.json 2> gpurun_out/k/kt.err
python tools/rocpd_stats.py $(find gpurun_out/k/kt -name "*.db" | head -1) > gpurun_out/k4_bench_kernel_stats.md 2>> gpurun_out/k/kt.err
head -8 gpurun_out/k4_bench_kernel_stats.md
tail -c 400 gpurun_out/k/bench_prof.json | head -c 400; echo
bash tools/pmc_prefill.sh > gpurun_out/k5_prefill_pmc_raw.txt 2>&1
grep -c prefill gpurun_out/k5_prefill_pmc_raw.txt
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $P -d gpurun_out/k/pmc_$P -- python tools/kbench.py --only "yi6b" --variants 0 > /dev/null 2> gpurun_out/k/pmc_$P.err
done
python - > gpurun_out/k6_hbm_pmc_raw.txt 2>&1 <<'PY'
import sqlite3, glob
for P in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/k/pmc_%s/**/*.db" % P, recursive=True)
    if not f:
        print(P, "no database"); continue
    db = sqlite3.connect(f[0])
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    g = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = "select substr(kernel_name,1,62), counter_name, %s, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%%vattn%%' group by kernel_name, counter_name%s order by kernel_name" % (g or "0", (", " + g) if g else "")
    for r in db.execute(q):
        print("%-62s %-10s grid_threads=%-9s n=%-3d per-dispatch mean %.4g  min %.4g  max %.4g" % r)
PY
cat gpurun_out/k6_hbm_pmc_raw.txt
timeout 300 python tools/kbench.py > gpurun_out/k7_kbench.txt 2>&1
grep -v amdgpu gpurun_out/k7_kbench.txt | head -60
(echo "#### decode, one 16-head block per workgroup (variant 128) vs the default (two blocks when G > 16)"; timeout 120 python tools/kbench.py decode --only "G32,G64" --variant 128; timeout 120 python tools/kbench.py decode --only "G32,G64" --variant 0) 2>&1 | grep -v amdgpu > gpurun_out/k8_kbench_decode_nb.txt
cat gpurun_out/k8_kbench_decode_nb.txt
for n in 2 4 8; do
  timeout 400 python bench.py --rank-of $n --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/k9_bench_rank_of_$n.json 2> gpurun_out/k/rank_of_$n.err
  tail -c 300 gpurun_out/k9_bench_rank_of_$n.json | head -c 300; echo
done
find gpurun_out/k -name "*.db" -size +20M -delete
du -sh gpurun_out/k
