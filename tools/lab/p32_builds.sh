#!/bin/bash
# LAB: schedule variants of prefill32_kernel (tools/lab/csrc/prefill32_lab.hip) as whole lab libraries (build/p32/libvattn_lab_<tag>.so): the
# lab objects of build/obj_lab with prefill32_lab.hip recompiled under -DP32_SCHED=NA,RING,MS,BJ,D0,DS; copy one over
# tools/lab/libvattn_lab.so and time `tools/kbench.py prefill --variant 6`.   usage: tools/lab/p32_builds.sh tag=NA,RING,MS,BJ,D0,DS ...
cd "$(dirname "$0")/../.."
mkdir -p build/p32
for spec in "$@"; do
    tag=${spec%%=*}; sched=${spec#*=}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-inline-asm -mllvm -pragma-unroll-threshold=1000000 \
        -Iinclude -Ivattention_amd/csrc -DVATTN_LAB "-DP32_SCHED=$sched" -Rpass-analysis=kernel-resource-usage -c tools/lab/csrc/prefill32_lab.hip -o build/p32/p32_$tag.o 2> build/p32/p32_$tag.log || { tail -20 build/p32/p32_$tag.log; exit 1; }
    grep -h "VGPRs:\|VGPRs Spill\|ScratchSize" build/p32/p32_$tag.log | sed 's/.*remark: [^ ]* *//; s/\[-R.*//' | sort | uniq -c | tr '\n' ';'; echo " <- $tag ($sched)"
    objs=$(ls build/obj_lab/*.o | grep -v prefill32_lab)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -Wl,-Bsymbolic $objs build/p32/p32_$tag.o -o build/p32/libvattn_lab_$tag.so
done
