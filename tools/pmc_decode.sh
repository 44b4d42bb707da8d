#!/bin/bash
# SQ / L2 counters of decode_kernel on the configs[1] decode shape (batch 16 at 32 k, Yi-6B heads): separate --pmc passes (never combined
# with API traces), summarised per kernel and grid size from the rocpd databases.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
P2="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P -d gpurun_out/pmcd_$i -- python tools/kbench.py decode --only "B16@32k" > /dev/null 2> gpurun_out/pmcd_$i.err
done
python - <<'PY'
import sqlite3, glob
for i in (1, 2):
    f = glob.glob("gpurun_out/pmcd_%d/**/*.db" % i, recursive=True)
    if not f:
        print("pass %d: no database" % i); continue
    db = sqlite3.connect(f[0])
    for r in db.execute("select substr(kernel_name,1,60), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%vattn%' group by kernel_name, counter_name order by kernel_name, counter_name"):
        print("%-60s %-26s n=%d per-dispatch %.4g" % r)
PY
rm -rf gpurun_out/pmcd_1 gpurun_out/pmcd_2
