#!/bin/bash
# Round 5, GPU call 1: (a) clock / power telemetry sources on the box, under a matrix-bound load; (b) cold vs warm pool — kernel time vs
# launch gap of the prefill launches while the mapper thread creates the pool's handles (rocprofv3 --kernel-trace --hip-trace, no
# counters); (c) two waves per SIMD (8 waves x 32 rows, prefill_kernel) against one wave per SIMD (prefill64) on the configs[1] prompt,
# times alternating + one --pmc pass for the matrix-pipe duty of both.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c1; mkdir -p $O; R=$PWD
echo "== (a) telemetry =="
python -m vattention_amd.telemetry --probe 2>&1 | tail -4 | tee $O/telemetry_probe.txt
timeout 300 python - > $O/telemetry_load.txt 2>&1 <<'PY'
import sys, time, subprocess
sys.path.insert(0, ".")
from vattention_amd.telemetry import Sampler
t0 = time.time()
with Sampler(0, 0.05) as s:
    subprocess.run([sys.executable, "tools/kbench.py", "prefill", "--variant", "0", "--only", "yi6b whole"], check=False)
    t1 = time.time()
print("whole run", s.window(t0, t1))
print("last 2 s ", s.window(t1 - 2.0, t1))
PY
tail -5 $O/telemetry_load.txt
echo "== (b) cold pool =="
CMD="python tools/dynamic_stress.py --model llama-3-8b --page-kib 8192 --megacache --requests 96 --batch 256 --passes 2"
rm -rf /tmp/cp; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace -d /tmp/cp -- bash -c "cd $R && $CMD" ) > $O/cold_pool_run.log 2>&1
DB=$(find /tmp/cp -name "*.db" | head -1)
python tools/cold_pool_trace.py $DB > $O/cold_pool_trace.txt 2>&1; cat $O/cold_pool_trace.txt
python tools/mapper_overlap.py $DB > $O/cold_pool_overlap.txt 2>&1; tail -12 $O/cold_pool_overlap.txt
grep -v amdgpu.ids $O/cold_pool_run.log | tail -3 | cut -c1-600
rm -rf /tmp/cp
echo "== (c) two waves per SIMD vs one =="
for i in 1 2; do
  for v in 2 14; do timeout 200 python tools/kbench.py prefill --variant $v --only "yi6b whole,yi6b chunk4k@28k" 2>&1 | grep -v "^==\|amdgpu.ids"; done
done | tee $O/two_wave_ab.txt
for v in 2 14; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/pmc_v$v -- python tools/kbench.py prefill --variant $v --only "yi6b whole" > /dev/null 2> $O/pmc_v$v.err
  echo "variant $v"; python tools/pmc_summary.py /tmp/pmc_v$v prefill
done 2>&1 | tee $O/two_wave_pmc.txt
