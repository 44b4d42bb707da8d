# per-kernel durations (rocprofv3 --kernel-trace --stats) of the stream decode path and the legacy grid on a few shapes
cd /tmp && export TMPDIR=/tmp
for v in 0 524288; do
  rm -rf /tmp/prof_ds_$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ds_$v -o ds -- python $GRAFT_REPO_ROOT/tools/kbench.py decode --variant $v --only "$1" > /tmp/prof_ds_$v.log 2>&1
  echo "== variant $v"; grep "splits=" /tmp/prof_ds_$v.log
  python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/prof_ds_$v -name "*.db" | head -1) | grep "decode\|combine" | cut -c1-200
done
