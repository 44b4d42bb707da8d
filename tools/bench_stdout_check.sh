#!/bin/bash
# bench.py's stdout must carry ONE JSON line (the capacity leg provokes the page manager's OOM state dump, which the library prints on fd 1).
cd "$(dirname "$0")/.."
timeout 600 python bench.py --leg capacity > /tmp/cap.out 2> /tmp/cap.err; echo "rc=$?"
echo "stdout lines: $(wc -l < /tmp/cap.out), bytes: $(wc -c < /tmp/cap.out)"; head -c 600 /tmp/cap.out; echo
echo "stderr lines: $(wc -l < /tmp/cap.err)"; grep -c "reqId" /tmp/cap.err
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/b.out 2> /tmp/b.err; echo "rc=$?"
echo "stdout lines: $(wc -l < /tmp/b.out), bytes: $(wc -c < /tmp/b.out)"; python3 -c "
import json; d=json.loads(open('/tmp/b.out').read()); print(d['value'], d['roofline']['frac'], d['roofline']['other']['decode']['frac'], sorted(d))"
