#!/bin/bash
# round 6, GPU call 5: the price of ONE more instruction per MFMA group, by class (64 per tile step; results unchanged)
cd "$(dirname "$0")/../.."
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r06c5; mkdir -p $O
bash tools/lab/pmc_p64_variants.sh $O/pmc_p64_price_list2.txt "5" "4 7 8 9 10 11"
