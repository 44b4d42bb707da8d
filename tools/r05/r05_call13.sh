#!/bin/bash
# Round 5, call 13 (the last two GPU minutes): the planner's pricing re-implemented on the host (identical lists, a tenth of the time) on the
# TP8-rank leg — at stride 8, where layer 0's event pair (host planning inside) is 1 of 10 timed launches, then at the product's stride 7.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c13; mkdir -p $O
leg() { timeout 75 python bench.py --leg dynamic_tp8_rank "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dynamic_tp8_rank']
w=d['warm_pool_pass']
f=lambda r: '%.4f (%.4f ms x %d timed of %d)' % (r['frac'], r['ms_per_launch'], r['launches_timed'], r['launches'])
print('  fresh pool %.0f tok/s prefill %s decode %.4f' % (d['tokens_per_s'], f(d['roofline_prefill']), d['roofline_decode']['frac']))
print('  warm pool  %.0f tok/s prefill %s decode %.4f' % (w['tokens_per_s'], f(w['roofline_prefill']), w['roofline_decode']['frac']))
"; }
{
echo "== fast planner pricing, stride 8"; leg --timer-every 8
echo "== fast planner pricing, stride 7"; leg --timer-every 7
} 2>&1 | tee $O/fast_planner_leg.txt
