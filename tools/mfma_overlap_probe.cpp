// Can a gfx950 SIMD keep its matrix pipe busy while VALU work issues next to it?  Explicit instruction streams
// (asm volatile: nothing is merged or reordered by the compiler).
//   same-wave   : every wave runs  MFMA, n x VALU, MFMA, n x VALU ...        (what a fused softmax/attention wave does)
//   split-wave  : even waves run only MFMAs, odd waves only VALU             (what a ping-pong schedule relies on)
// Variants: accumulators in arch VGPRs or AccVGPRs, 1/2/4 waves per SIMD, s_setprio on the matrix waves,
// v_fma_f32 / v_pk_fma_f32 / v_exp_f32 as the VALU op.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.cpp -o mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define VPK(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p1), "v"(p2))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// KIND: 0 v_fma, 1 v_pk_fma, 2 v_exp.  ACC: 0 VGPR, 1 AGPR.  SPLIT: 0 same-wave, 1 even=MFMA odd=VALU, 2 = split + setprio.
// NV = VALU instructions per MFMA (same-wave) or per MFMA-time slot of the partner wave (split).
template <int NV, int KIND, int ACC, int SPLIT, int THREADS, int MINB>
__global__ __launch_bounds__(THREADS, MINB) void k(float* out, int iters, float seed, int do_mfma, int do_valu) {
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float x[8]; f32x2 y[8];
    for (int j = 0; j < 8; j++) { x[j] = seed + j; y[j][0] = seed + j; y[j][1] = seed - j; }
    float c1 = 1.0001f, c2 = 0.5f; f32x2 p1 = {1.0001f, 0.9999f}, p2 = {0.5f, 0.25f};
    const int wave = threadIdx.x >> 6;
    // waves of a workgroup go round-robin over the 4 SIMDs: wave w -> SIMD w & 3, so roles are split on bit 2
    // (SPLIT 3 splits on bit 0 instead: matrix waves on SIMDs 0/2, VALU waves on 1/3 - the control experiment)
    const int role = SPLIT == 3 ? (wave & 1) : ((wave >> 2) & 1);
    const bool mf = SPLIT ? (role == 0) : true;
    const bool va = SPLIT ? (role == 1) : true;
    if (SPLIT == 2 && mf) __builtin_amdgcn_s_setprio(3);
#define VAL(j) do { if (KIND == 0) VFMA(x[(j) & 7]); else if (KIND == 1) VPK(y[(j) & 7]); else VEXP(x[(j) & 7]); } while (0)
    if (!SPLIT) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (do_mfma) {
                    if (ACC) { if (c == 0) MFMA_A(acc0); else if (c == 1) MFMA_A(acc1); else if (c == 2) MFMA_A(acc2); else MFMA_A(acc3); }
                    else     { if (c == 0) MFMA_V(acc0); else if (c == 1) MFMA_V(acc1); else if (c == 2) MFMA_V(acc2); else MFMA_V(acc3); }
                }
                if (do_valu) {
#pragma unroll
                    for (int n = 0; n < NV; n++) VAL(c * NV + n);
                }
            }
        }
    } else if (mf) {
        if (do_mfma) for (int i = 0; i < iters; i++) {
            if (ACC) { MFMA_A(acc0); MFMA_A(acc1); MFMA_A(acc2); MFMA_A(acc3); }
            else     { MFMA_V(acc0); MFMA_V(acc1); MFMA_V(acc2); MFMA_V(acc3); }
        }
    } else if (va) {
        if (do_valu) for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int n = 0; n < 4 * NV; n++) VAL(n);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    for (int j = 0; j < 8; j++) s += x[j] + y[j][0] + y[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int ACC, int SPLIT, int THREADS, int MINB>
void run(const char* name) {
    float* out; hipMalloc(&out, 512 * 1024 * 4);
    const int iters = 10000;
    const int grid = 256 * MINB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    for (int mode = 0; mode < 3; mode++) {       // 0: both, 1: matrix only, 2: VALU only
        const int dm = mode != 2, dv = mode != 1;
        k<NV, KIND, ACC, SPLIT, THREADS, MINB><<<grid, THREADS>>>(out, 100, 0.25f, dm, dv);
        hipEventRecord(e0);
        k<NV, KIND, ACC, SPLIT, THREADS, MINB><<<grid, THREADS>>>(out, iters, 0.25f, dm, dv);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double mwaves = (double)grid * (THREADS / 64) / (SPLIT ? 2 : 1);
    const double flops = mwaves * iters * 4.0 * 32768.0;
    printf("%-44s NV=%2d  both %7.3f ms (%6.1f TF)  mfma-only %7.3f ms (%6.1f TF)  valu-only %7.3f ms   both/max(parts) %.2f\n", name, NV,
           ms[0], flops / ms[0] / 1e9, ms[1], flops / ms[1] / 1e9, ms[2], ms[0] / (ms[1] > ms[2] ? ms[1] : ms[2]));
    hipFree(out);
}

int main() {
    // waves/SIMD sweep, matrix only is column 2
    run<4, 0, 0, 0, 256, 1>("same v_fma VGPRacc 1w/SIMD");
    run<4, 0, 0, 0, 512, 1>("same v_fma VGPRacc 2w/SIMD");
    run<4, 0, 0, 0, 512, 2>("same v_fma VGPRacc 4w/SIMD");
    run<4, 0, 1, 0, 512, 1>("same v_fma AGPRacc 2w/SIMD");
    run<4, 0, 1, 0, 512, 2>("same v_fma AGPRacc 4w/SIMD");
    run<2, 0, 0, 0, 512, 1>("same v_fma VGPRacc 2w/SIMD");
    run<6, 0, 0, 0, 512, 1>("same v_fma VGPRacc 2w/SIMD");
    run<7, 0, 0, 0, 512, 1>("same v_fma VGPRacc 2w/SIMD");
    run<8, 0, 0, 0, 512, 1>("same v_fma VGPRacc 2w/SIMD");
    run<6, 0, 1, 0, 512, 1>("same v_fma AGPRacc 2w/SIMD");
    run<8, 0, 1, 0, 512, 1>("same v_fma AGPRacc 2w/SIMD");
    run<4, 1, 0, 0, 512, 1>("same v_pk_fma VGPRacc 2w/SIMD");
    run<4, 1, 1, 0, 512, 1>("same v_pk_fma AGPRacc 2w/SIMD");
    run<2, 2, 0, 0, 512, 1>("same v_exp VGPRacc 2w/SIMD");
    run<2, 2, 1, 0, 512, 1>("same v_exp AGPRacc 2w/SIMD");
    // split: 2 waves/SIMD = one matrix wave + one VALU wave per SIMD (wave ids alternate, SIMD = wave & 3 -> use 4w/SIMD too)
    run<4, 0, 0, 1, 512, 1>("split v_fma VGPRacc 8 waves/CU");
    run<6, 0, 0, 1, 512, 1>("split v_fma VGPRacc 8 waves/CU");
    run<7, 0, 0, 1, 512, 1>("split v_fma VGPRacc 8 waves/CU");
    run<4, 0, 1, 1, 512, 1>("split v_fma AGPRacc 8 waves/CU");
    run<7, 0, 1, 1, 512, 1>("split v_fma AGPRacc 8 waves/CU");
    run<7, 0, 1, 2, 512, 1>("split+prio v_fma AGPRacc 8 waves/CU");
    run<7, 0, 0, 3, 512, 1>("split-by-bit0 (control) v_fma 8 waves/CU");
    run<4, 0, 0, 1, 1024, 1>("split v_fma VGPRacc 16 waves/CU");
    run<7, 0, 0, 1, 1024, 1>("split v_fma VGPRacc 16 waves/CU");
    run<7, 0, 1, 2, 1024, 1>("split+prio v_fma AGPRacc 16 waves/CU");
    run<2, 2, 0, 1, 1024, 1>("split v_exp VGPRacc 16 waves/CU");
    run<2, 2, 1, 2, 1024, 1>("split+prio v_exp AGPRacc 16 waves/CU");
    run<4, 1, 1, 2, 1024, 1>("split+prio v_pk_fma AGPRacc 16 waves/CU");
    return 0;
}
