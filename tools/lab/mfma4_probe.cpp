// LAB probe (round 6): what v_mfma_f32_4x4x4_16b_f16 does with A = ones — is D[i] of a lane the sum of THAT lane's four B values?
// build: hipcc --offload-arch=gfx950 -O2 tools/lab/mfma4_probe.cpp -o tools/lab/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int swap) {
    const int l = threadIdx.x;
    h4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    h4 b = {(_Float16)(l), (_Float16)(0.25f), (_Float16)(0.5f), (_Float16)(0.125f)};      // lane-local sum = l + 0.875
    if (swap == 2) { b[0] = (_Float16)1.f; b[1] = (_Float16)0.f; b[2] = (_Float16)0.f; b[3] = (_Float16)0.f; }
    if (swap == 3) { b[0] = (_Float16)0.f; b[1] = (_Float16)1.f; b[2] = (_Float16)0.f; b[3] = (_Float16)0.f; }
    if (swap == 4) { b[0] = (_Float16)0.f; b[1] = (_Float16)0.f; b[2] = (_Float16)1.f; b[3] = (_Float16)0.f; }
    if (swap == 5) { b[0] = (_Float16)0.f; b[1] = (_Float16)0.f; b[2] = (_Float16)0.f; b[3] = (_Float16)1.f; }
    f4 d = {0.f, 0.f, 0.f, 0.f};
    if (swap == 1) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(d) : "v"(b), "v"(ones));
    else asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(d) : "v"(ones), "v"(b));
    for (int i = 0; i < 4; i++) out[(swap * 64 + l) * 4 + i] = d[i];
}
int main() {
    float* o;
    hipMalloc(&o, 6 * 64 * 4 * 4);
    k<<<1, 64>>>(o, 0);
    for (int s = 1; s < 6; s++) k<<<1, 64>>>(o, s);
    float h[6 * 64 * 4];
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    for (int s = 0; s < 6; s++) {
        printf("case %d (0: ones as A, b = {l, .25, .5, .125}; 1: roles swapped; 2-5: ones as A, b = e_(case-2))\n", s);
        for (int l = 0; l < 6; l++) printf("  lane %2d (own sum %6.3f): D = %9.6f %9.6f %9.6f %9.6f\n", l, l + 0.875, h[(s * 64 + l) * 4], h[(s * 64 + l) * 4 + 1], h[(s * 64 + l) * 4 + 2], h[(s * 64 + l) * 4 + 3]);
    }
    return 0;
}
