#!/bin/bash
# LAB (round 6): one SQ counter pass over the prefill64 builds behind variant bits 28-30 on the configs[1] prompt — product (0), V^T pre-read
# (1), and the issue-budget ablations 4-7 (WRONG results: instructions deleted) — cycles per tile step, matrix-pipe duty, issue / wait split.
# usage: bash tools/lab/pmc_p64_variants.sh [outfile] [r6 selectors, default "1 4 5 6 7"]
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=${1:-gpurun_out/pmc_p64_variants.txt}
V=14
for s in ${2:-1 4 5 6 7}; do V="$V,$((14 + (s << 28)))"; done
# $3: sub-selector of build 7 (KBENCH_LAB_SUB): one pass per value, appended
rm -rf /tmp/pmcv /tmp/pmcv_sub*
CTRS="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
( cd /tmp && timeout 600 rocprofv3 --pmc $CTRS -d /tmp/pmcv -- python $OLDPWD/tools/kbench.py prefill --variants $V --only "yi6b whole" ) > /tmp/pmcv.log 2>&1
for sub in $3; do
  ( cd /tmp && KBENCH_LAB_SUB=$sub timeout 300 rocprofv3 --pmc $CTRS -d /tmp/pmcv_sub$sub -- python $OLDPWD/tools/kbench.py prefill --variants $((14 + (7 << 28))) --only "yi6b whole" ) >> /tmp/pmcv.log 2>&1
done
python - > $OUT <<'PY'
import sqlite3, glob, re
rows = {}
for f in glob.glob("/tmp/pmcv*/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, ctr, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%prefill64_kernel%' group by kernel_name, counter_name"):
        rows.setdefault(name, {})[ctr] = avg
label = {"ELi9ELi3ELi517ELi0E": "pre-read + scalar scale + DMA distances in soffset",
         "ELi9ELi3ELi261ELi0E": "pre-read + scalar scale + row sums by 4x4x4 MFMA",
         "ELi5ELi1E": "+64 s_nop 0 per tile", "ELi5ELi2E": "+64 s_mov (SALU) per tile", "ELi5ELi3E": "+64 s_waitcnt (no wait) per tile", "ELi5ELi4E": "+64 v_mov per tile",
         "ELi5ELi5E": "+64 v_exp per tile", "ELi5ELi6E": "+64 v_max3 per tile", "ELi5ELi7E": "+64 v_pk_fma_f32 per tile", "ELi5ELi8E": "+64 v_pk_add_f32 per tile",
         "ELi5ELi9E": "+64 v_add_f32 per tile", "ELi5ELi10E": "+64 v_cvt_pk_f16_f32 per tile", "ELi5ELi11E": "+64 v_pk_mul_f32 per tile",
         "ELi9ELi3ELi0ELi0E": "product (library of this tree)", "ELi9ELi3ELi1ELi0E": "V^T pre-read",
         "ELi9ELi3ELi5ELi0E": "pre-read + scalar scale", "ELi9ELi3ELi9ELi0E": "pre-read + order M,E,A",
         "ELi9ELi3ELi13ELi0E": "pre-read + scalar scale + M,E,A",
         "ELi9ELi3ELi17ELi0E": "fma -> v_mul VOP2 4 B (wrong)", "ELi9ELi3ELi33ELi0E": "fma -> v_mul VOP3 8 B (wrong)",
         "ELi9ELi3ELi65ELi0E": "v_mov + v_fmac (4 B each), exact", "ELi9ELi3ELi193ELi0E": "v_mov + v_fmac, 2 v_max per max3, exact",
         "ELi2176E": "no v_fma (wrong)", "ELi6272E": "no v_fma, no v_add (wrong)", "ELi130E": "no fma/exp/add (wrong)", "ELi166E": "MFMA + reads + DMA only (wrong)"}
print("%-36s %9s %9s %8s %8s %8s %8s %8s %8s" % ("build", "VALU/MFMA", "cyc/tile", "duty", "active", "wait_ins", "wait_any", "GUI/8", "MFMA"))
for name, c in sorted(rows.items()):
    lab = next((v for k, v in label.items() if k in name), None) or re.sub(r"^.*prefill64_kernelI", "", name)[:34]
    mf = c["SQ_INSTS_MFMA"]
    tiles = mf / 64.0                                  # wave tile steps
    cyc_tile = 4.0 * c["SQ_WAVE_CYCLES"] / tiles       # SQ_WAVE_CYCLES counts quad-cycles
    duty = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
    w = c["SQ_WAVE_CYCLES"]
    print("%-36s %9.2f %9.0f %8.3f %8.3f %8.3f %8.3f %8.3g %8.3g" % (lab, (c["SQ_INSTS_VALU"] - mf) / mf, cyc_tile, duty, c["SQ_ACTIVE_INST_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w,
                                                                   c["SQ_WAIT_ANY"] / w, c["GRBM_GUI_ACTIVE"] / 8, mf))
PY
grep -v amdgpu.ids /tmp/pmcv.log | grep "yi6b whole" >> $OUT
cat $OUT
