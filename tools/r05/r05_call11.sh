#!/bin/bash
# Round 5, call 11 (the last GPU minutes): (1) the prefill planner's finer cut candidates IN SITU — `bench.py --leg dynamic_tp8_rank` with the
# working tree's library (A) against the library of commit e4c1961 (B: the planner before them; build/base via tools/build_base.py e4c1961),
# alternating on one box; (2) parity of the decode lab forms (XCD-consecutive ranges, hand-over in the XCD's L2); (3) their timing.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c11; mkdir -p $O
cp vattention_amd/libvattn_amd.so /tmp/new.so
leg() { timeout 150 python bench.py --leg dynamic_tp8_rank 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dynamic_tp8_rank']
w=d['warm_pool_pass']
print('  cold %.0f tok/s prefill %s decode %s | warm %.0f tok/s prefill %s decode %s' % (d['tokens_per_s'], d['roofline_prefill'].get('frac'), d['roofline_decode'].get('frac'), w['tokens_per_s'], w['roofline_prefill'].get('frac'), w['roofline_decode'].get('frac')))
print('  warm prefill detail', json.dumps(w['roofline_prefill']))
"; }
{
for i in 1 2; do
  cp /tmp/new.so vattention_amd/libvattn_amd.so; echo "== A (working tree: finer cut candidates)"; leg
  cp build/base/libvattn_amd.so vattention_amd/libvattn_amd.so; echo "== B (e4c1961: the planner before them)"; leg
done
} 2>&1 | tee $O/planner_insitu_ab.txt
cp /tmp/new.so vattention_amd/libvattn_amd.so
timeout 240 python -m pytest tests/test_gpu_attention.py -m gpu -q -x --timeout 200 -k "xcd or test_decode_stream_plan" > $O/xcd_tests.log 2>&1; echo "xcd tests rc=$?" | tee -a $O/xcd_tests.log; tail -3 $O/xcd_tests.log | cut -c1-300
timeout 200 python tools/decode_xcd_probe.py 3 2>&1 | grep -v amdgpu.ids | tee $O/decode_xcd_probe.txt
