#!/bin/bash
# Round 5, call 12 (the last GPU minutes): the bench line with the replay legs' event stride coprime with the layer count (bench.py
# TIMER_EVERY), then the SAME library and box under stride 8 (rounds 4-5) and 7, and the persistent / per-piece prefill policies re-read with
# the unbiased stride on the TP8-rank leg.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r05c12; mkdir -p $O
timeout 420 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
leg() { timeout 120 python bench.py --leg dynamic_tp8_rank "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dynamic_tp8_rank']
w=d['warm_pool_pass']
f=lambda r: '%.4f (%.4f ms x %d timed of %d)' % (r['frac'], r['ms_per_launch'], r['launches_timed'], r['launches'])
print('  fresh pool %.0f tok/s prefill %s decode %.4f' % (d['tokens_per_s'], f(d['roofline_prefill']), d['roofline_decode']['frac']))
print('  warm pool  %.0f tok/s prefill %s decode %.4f' % (w['tokens_per_s'], f(w['roofline_prefill']), w['roofline_decode']['frac']))
"; }
{
echo "== stride 8 (rounds 4-5: layer 0 of every iteration sampled, seven of eight layers never)"; leg --timer-every 8
echo "== stride 7 (coprime with 80 layers: every layer equally often)"; leg --timer-every 7
echo "== stride 7, EVERY work list through persistent workgroups"; leg --timer-every 7 --persistent-prefill
echo "== stride 7, NO persistent workgroups"; leg --timer-every 7 --per-piece-prefill
} 2>&1 | tee $O/timer_stride.txt
