#!/bin/bash
# SQ counters of the default prefill kernel on the configs[1] prompt (32 702 tokens, Yi-6B heads): three separate --pmc passes
# (never combined with API traces), summarised per kernel from the rocpd databases.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P2="SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P -d gpurun_out/pmcp_$i -- python tools/kbench.py prefill --variants 0 --only "yi6b whole" > /dev/null 2> gpurun_out/pmcp_$i.err
done
python - <<'PY'
import sqlite3, glob
for i in (1, 2, 3):
    f = glob.glob("gpurun_out/pmcp_%d/**/*.db" % i, recursive=True)
    if not f:
        print("pass %d: no database" % i); continue
    db = sqlite3.connect(f[0])
    for r in db.execute("select substr(kernel_name,1,60), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%prefill%' group by kernel_name, counter_name order by counter_name"):
        print("%-60s %-30s n=%d per-dispatch %.4g" % r)
PY
rm -rf gpurun_out/pmcp_1 gpurun_out/pmcp_2 gpurun_out/pmcp_3
