#!/bin/bash
# LAB ONLY (round 4, prefill64 energy pass (ii)): two copies of the PRODUCT library that differ only in prefill64_kernels' device code —
#   build/asm_ab/libvattn_amd_asis.so   the compiler's assembly, reassembled unchanged (control for the pipeline / code placement)
#   build/asm_ab/libvattn_amd_nonop.so  the same assembly minus the `s_nop 0` between inline-asm statements (tools/lab/strip_asm_nops.py)
# Needs build/obj/*.o of the product build (python vattention_amd/build.py).  Timed by tools/p64_asm_ab.sh on the GPU.
set -e
cd "$(dirname "$0")/../.."
R=$PWD; L=/opt/rocm/lib/llvm/bin; W=build/asm_ab; mkdir -p $W
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-value -Wno-inline-asm"
SRC=vattention_amd/csrc/prefill64_kernels.hip
/opt/rocm/bin/hipcc $F --offload-device-only -S $SRC -o $W/dev.s 2>/dev/null
python3 tools/lab/strip_asm_nops.py $W/dev.s $W/dev_nonop.s
cp $W/dev.s $W/dev_asis.s
for v in asis nonop; do
    $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_$v.s -o $W/dev_$v.o
    $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $W/dev_$v.o -o $W/dev_$v.out
    $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev_$v.out -output=$W/dev_$v.hipfb
    /opt/rocm/bin/hipcc $F --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev_$v.hipfb -c $SRC -o $W/p64_$v.o 2>/dev/null
    objs=$(ls build/obj/*.o | grep -v prefill64_kernels)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -Wl,-Bsymbolic $objs $W/p64_$v.o -o $W/libvattn_amd_$v.so
    echo "$v: $($L/llvm-objdump -d $W/dev_$v.out | grep -c 's_nop 0') s_nop 0 in the device code"
done
