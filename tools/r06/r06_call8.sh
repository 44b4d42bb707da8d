#!/bin/bash
# round 6, GPU call 8: the product library with the round-6 prefill64 schedule — parity (prefill, persistent queues, fuzz, full size),
# then A (working tree) / B (round 5's library, build/base) / A on one box
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_parity.py tests/test_gpu_prefill_persistent.py tests/test_gpu_full_size_properties.py tests/test_gpu_docstring_pins.py tests/test_gpu_rope_fusion.py tests/test_gpu_wrapper_golden.py -m gpu -q --timeout 600 -x \
    -k "not decode" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
bash tools/p64_ab.sh notests > $O/ab.txt 2>&1; cat $O/ab.txt | grep -v amdgpu.ids
