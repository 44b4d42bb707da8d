cd /root/repo; export TMPDIR=/tmp
for sc in 1 0; do
  KBENCH_DATA_SCALE=$sc timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/clk_$sc -- python tools/kbench.py prefill --variants 0 --only "yi6b whole" > /dev/null 2> gpurun_out/clk_$sc.err
done
python - <<'PY'
import sqlite3, glob
for sc in (1, 0):
    db = sqlite3.connect(glob.glob("gpurun_out/clk_%d/**/*.db" % sc, recursive=True)[0])
    rows = db.execute("select c.value, c.duration, k.start, k.end from counters_collection c join kernels k on k.dispatch_id = c.dispatch_id where c.kernel_name like '%prefill_kernel%' and c.counter_name='GRBM_GUI_ACTIVE'").fetchall()
    if not rows:
        rows = db.execute("select value, duration, start, end from counters_collection where kernel_name like '%prefill_kernel%' and counter_name='GRBM_GUI_ACTIVE'").fetchall()
    for v, d, s, e in rows[:6]:
        dur = (e - s) if e and s else d
        print("data scale %d: GRBM_GUI_ACTIVE %.4g cycles over %.3f ms -> %.0f MHz" % (sc, v, dur / 1e6, v / (dur / 1e3) if dur else 0))
PY
rm -rf gpurun_out/clk_1 gpurun_out/clk_0
