set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/prof2
timeout 280 python bench.py --steps 2 --warmup 1 > gpurun_out/prof2/bench_n1.json 2> gpurun_out/prof2/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof2/kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof2/bench_prof.json 2> gpurun_out/prof2/kt.err
python tools/rocpd_stats.py $(find gpurun_out/prof2/kt -name "*.db" | head -1) > gpurun_out/prof2/bench_kernel_stats.md 2>> gpurun_out/prof2/kt.err
timeout 200 python tools/kbench.py > gpurun_out/prof2/kbench.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof2/pmc_fetch -- python tools/kbench.py --only "yi6b" --variants 0 > /dev/null 2> gpurun_out/prof2/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof2/pmc_write -- python tools/kbench.py --only "yi6b" --variants 0 > /dev/null 2> gpurun_out/prof2/pmc_write.err
(python tools/pmc_summary.py gpurun_out/prof2/pmc_fetch; python tools/pmc_summary.py gpurun_out/prof2/pmc_write) > gpurun_out/prof2/hbm_pmc_raw.txt 2>&1
find gpurun_out/prof2 -name "*.db" -size +20M -delete
du -sh gpurun_out/prof2
