#!/bin/bash
# HIP VMM calls against kernel execution on the final page manager: rocprofv3 --kernel-trace --hip-trace (no counters) of a cold
# 64-request dynamic replay, summarised by tools/mapper_overlap.py.  usage: bash tools/mapper_overlap.sh > profiles/rNN_mapper_overlap.md
cd "$(dirname "$0")/.."; export TMPDIR=/tmp PYTHONUNBUFFERED=1
CMD="python tools/dynamic_stress.py --layers 16 --pool-gib 60 --requests 64 --batch 64 --page-kib 8192 --megacache --passes 1"
R=$PWD
rm -rf /tmp/mo; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace -d /tmp/mo -- bash -c "cd $R && $CMD" ) > /tmp/mo.log 2>&1
echo "# HIP VMM calls vs kernel execution, round 4 (rocprofv3 --kernel-trace --hip-trace, 64-request dynamic replay, cold pool, megacache 8 MiB pages, 16 layers)"
echo; echo "Command: \`rocprofv3 --kernel-trace --hip-trace -- $CMD\`, summarised by \`tools/mapper_overlap.py\`."; echo
python tools/mapper_overlap.py $(find /tmp/mo -name "*.db" | head -1)
echo; echo '```'; grep -v "amdgpu.ids\|^\[vattn\] warning" /tmp/mo.log | tail -6; echo '```'
