#!/bin/bash
# The round's profiling evidence in one go (bash tools/prof_round.sh [outdir]): rocprofv3 kernel statistics of the bench workload
# (whose average launch durations must agree with the HIP-event timings bench.py reports), HBM traffic of the two dominant kernels
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, never combined with API traces), SQ / L2 counters of both (tools/pmc_*.sh).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=${1:-gpurun_out/prof6}
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-dynamic --no-cpu-baseline > $O/bench_prof.json 2> $O/kt.err
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/bench_kernel_stats.md 2>> $O/kt.err
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python tools/kbench.py --only "yi6b whole,yi6b B16@32k" --variants 0 > /dev/null 2> $O/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python tools/kbench.py --only "yi6b whole,yi6b B16@32k" --variants 0 > /dev/null 2> $O/pmc_write.err
(python tools/pmc_summary.py $O/pmc_fetch vattn; python tools/pmc_summary.py $O/pmc_write vattn) > $O/hbm_pmc_raw.txt 2>&1
bash tools/pmc_decode.sh > $O/decode_pmc_raw.txt 2>&1
bash tools/pmc_prefill.sh > $O/prefill_pmc_raw.txt 2>&1
python tools/traffic_json.py $O/hbm_pmc_raw.txt $O/prefill_pmc_raw.txt $O/bench_kernel_stats.md > $O/traffic.json 2>> $O/kt.err
find $O -name "*.db" -delete
rm -rf $O/kt $O/pmc_fetch $O/pmc_write
tail -c 600 $O/bench_prof.json; head -30 $O/bench_kernel_stats.md; cat $O/hbm_pmc_raw.txt | tail -12
