#!/bin/bash
# LAB (round 6): one SQ counter pass over prefill64_kernel (variant 14) and prefill32_kernel (variant 6) on the configs[1] prompt:
# busy cycles of the chip, matrix-pipe duty, VALU per MFMA, LDS activity and bank conflicts.   usage: bash tools/lab/pmc_p32.sh [outfile] [workload]
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=${1:-gpurun_out/pmc_p32.txt}
WL=${2:-yi6b whole}
rm -rf /tmp/pmc32a /tmp/pmc32b
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pmc32a -- python $OLDPWD/tools/kbench.py prefill --variants 14,6 --only "$WL" ) > /tmp/pmc32.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d /tmp/pmc32b -- python $OLDPWD/tools/kbench.py prefill --variants 14,6 --only "$WL" ) >> /tmp/pmc32.log 2>&1
python - > $OUT <<'PY'
import sqlite3, glob, re
rows = {}
for f in glob.glob("/tmp/pmc32*/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%prefill%_kernel%' group by kernel_name, counter_name"):
        rows.setdefault(re.sub(r"^.*(prefill\d+_kernel).*$", r"\1", name), {}).setdefault(ctr, []).append(avg)
for name, c in sorted(rows.items()):
    c = {k: sum(v) / len(v) for k, v in c.items()}
    mf, gui, w = c["SQ_INSTS_MFMA"], c["GRBM_GUI_ACTIVE"] / 8.0, c["SQ_WAVE_CYCLES"]
    print("%s: chip-busy cycles %.4g; cycles per 64 MFMAs per SIMD %.0f; matrix-pipe duty %.3f; VALU per MFMA %.2f; per wave: active %.3f wait_inst %.3f wait_any %.3f" % (
        name, gui, gui / (mf / 64.0 / 1024.0), c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui, (c["SQ_INSTS_VALU"] - mf) / mf, c["SQ_ACTIVE_INST_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w, c["SQ_WAIT_ANY"] / w))
    print("    LDS: instructions per MFMA %.2f; SQ_LDS_IDX_ACTIVE / (CU x busy cycles) %.3f; bank-conflict cycles / active %.4f; SQ_ACTIVE_INST_LDS per SIMD-cycle %.3f; wave-cycles waiting on LDS / wave cycles %s; SQ_ACTIVE_INST_VALU per SIMD-cycle %.3f" % (
        c["SQ_INSTS_LDS"] / mf, c["SQ_LDS_IDX_ACTIVE"] / 256.0 / gui, c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 4.0 * c["SQ_ACTIVE_INST_LDS"] / 1024.0 / gui,
        "%.3f" % (c["SQ_WAIT_INST_LDS"] / w), 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / gui))
    print("    raw:", {k: float("%.5g" % v) for k, v in sorted(c.items())})
PY
grep -v amdgpu.ids /tmp/pmc32.log | grep "$WL\|rror" >> $OUT
cat $OUT
