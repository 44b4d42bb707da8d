#!/bin/bash
# round 6, GPU call 11: decode — the drawn queue of fixed small pieces (lab, bit 21): parity, then A/B against the stream plan
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q --timeout 600 -k "test_decode_stream_plan and not table" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -8 $O/tests.log
timeout 1200 python tools/lab/decode_queue_ab.py 3 > $O/decode_queue_ab.txt 2>&1; echo "rc=$?" >> $O/decode_queue_ab.txt; grep -v amdgpu.ids $O/decode_queue_ab.txt
