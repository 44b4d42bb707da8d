#!/bin/bash
# round 6, GPU call 19: the bench line with roofline.other.power_ceiling / hbm_ceiling
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/c19
mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench_details.log
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c19/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "clock_mhz_mean", "power_w_mean")})
print({k: v for k, v in r.items() if k != "other"})
print(json.dumps(r["other"].get("power_ceiling")), json.dumps(r["other"].get("hbm_ceiling")))
print(r["other"]["prefill"], r["other"]["decode"])
PY
