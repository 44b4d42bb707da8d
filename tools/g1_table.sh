#!/bin/bash
# g1: what per-layer pages of 64 KiB / 256 KiB / 2 MiB cost on this platform, and BASELINE's configs[2] / configs[4]
# run AS WRITTEN at the pool size whose handles can still be created up front.  usage: bash tools/g1_table.sh > profiles/rNN_g1_small_pages.md
cd "$(dirname "$0")/.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "## hipMemCreate against the number of live handles, per page size (tools/vmm_scale_probe.cpp)"; echo '```'
for page in 65536 262144 2097152; do echo "page $page bytes"; timeout 120 tools/vmm_scale_probe $page 50000 30000 2>&1 | grep "live handles"; done
echo '```'
run() { echo; echo "## $1"; echo '```'; shift; timeout 600 python tools/dynamic_stress.py "$@" 2>&1 | grep -v "amdgpu.ids\|^\[vattn\] warning" | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        print(line[:300]); continue
    d = json.loads(line)
    keep = ('pass', 'requests', 'tokens', 'seconds', 'tokens_per_s', 'peak_running', 'tokens_per_page', 'kv_live_over_needed_at_peak', 'kv_live_over_mapped_mean', 'handles_created', 'create_ms',
            'map_calls', 'unmap_calls', 'sync_map_ms', 'async_map_ms', 'external_fragmentation_max', 'page_kib', 'megacache', 'pool_gib', 'layers', 'model', 'tp')
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in keep}))
"; echo '```'; }
run "configs[4] as written: llama-3-70b TP8 rank (80 layers x 1 kv head), 256 KiB per-layer pages, pool 9.7 GiB = 39 680 handles (created inside reserve), 48 requests, pass 1 fresh pool / pass 2 warm" \
    --model llama-3-70b --tp 8 --page-kib 256 --pool-gib 9.7 --requests 48 --batch 256 --passes 2
run "configs[2] as written: llama-3-8b TP1 (32 layers x 8 kv heads), 64 KiB per-layer pages (32 tokens per page), pool 2.44 GiB = 39 936 handles, 3 requests" \
    --model llama-3-8b --page-kib 64 --pool-gib 2.44 --requests 3 --batch 256 --passes 2
run "beyond the whole-pool window: the same TP8 rank, 256 KiB pages, pool 19.5 GiB = 79 872 handles (lazy: 4 096-handle window ahead of demand), 96 requests" \
    --model llama-3-70b --tp 8 --page-kib 256 --pool-gib 19.5 --requests 96 --batch 256 --passes 2
