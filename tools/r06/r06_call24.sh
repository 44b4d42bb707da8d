#!/bin/bash
# round 6, GPU call 24: bf16 decode without a scratch segment (a build of decode_stream_kernel without the fused-RoPE path for calls that ask for no
# rotation): decode / RoPE parity tests, then fp16 vs bf16 interleaved
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_rope_fusion.py tests/test_gpu_fuzz.py tests/test_gpu_docstring_pins.py -m "gpu and not lab" -q --timeout 600 -k "decode or rope or stream" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
for i in 1 2 3; do
  python tools/kbench.py decode --variants 0 --only "yi6b B16@32k,llama8b B64@8k,yi34b/tp2 B1@128k" 2>&1 | grep "B16@32k\|B64@8k\|B1@128k"
  python tools/kbench.py decode --bf16 --variants 0 --only "yi6b B16@32k,llama8b B64@8k,yi34b/tp2 B1@128k" 2>&1 | grep "B16@32k\|B64@8k\|B1@128k" | sed "s/^/bf16 /"
done | tee $O/dec_bf16_ab.txt
