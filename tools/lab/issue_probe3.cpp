// LAB probe (round 6): would WARP SPECIALISATION lift the prefill tile step off its VALU-issue bound?  (gfx950)
// prefill64_kernel runs one wave per SIMD that issues everything: per tile 64 MFMAs + 284 VALU + 48 LDS reads, and is bound by what that one wave
// can issue to the vector ALU (7.3 cycles per VALU instruction, profiles/r06_p64_price_list.txt).  A partner wave's VALU is nearly free for a wave
// (MI355X_MICROARCH: two waves per SIMD), so: split the step by ROLE — wave A scores and exponentiates (the S^T MFMAs + all softmax VALU, P written to
// LDS), wave B multiplies (the P.V MFMAs, V^T / P fragment reads, the DMA stream).  This probe times the instruction streams only (no data flow):
//   mode 0  one wave per SIMD: per 2 MFMAs the whole mix (2 fma, 2 exp, 2 add, 1 cvt_pk, 1 max3, 1/2 mov; 1/2 ds_read_b128, 1 ds_read_b64_tr)
//   mode 1  two waves per SIMD, each the whole mix (symmetric split: every fragment read feeds one MFMA, so twice the LDS reads per MFMA)
//   mode 3  as mode 1 with the LDS reads a 32-row wave really needs: every fragment feeds ONE MFMA (1/2 ds_read_b128 + 1 ds_read_b64_tr per MFMA)
//   mode 2  two waves per SIMD by ROLE: A = {1 MFMA + the VALU mix of two MFMAs + 1/2 ds_read_b128 + 1/4 ds_write_b128},
//                                      B = {1 MFMA + 1 ds_read_b64_tr + 1/4 ds_read_b128}, one s_barrier per 64 MFMAs (a tile)
// Operands are pseudo-random (power as on real data).  Output: ns per MFMA per SIMD, whole chip and 8 CUs.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/issue_probe3 tools/lab/issue_probe3.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct Regs { float x[16]; float k[4]; float y, y2, c; f4 s4; f2 s2; };

__device__ __forceinline__ void valu_pair(int q, Regs& r, int odd = 0) {      // the VALU work of TWO MFMAs (8.5 instructions) in two slices
    // no instruction reads a result younger than a whole pair (as in the kernel, where a score's fma / exp / add sit a group apart)
    if (q == 0) {
        asm volatile("v_fma_f32 %0, %2, %3, %3\n\tv_fma_f32 %1, %2, %3, %3" : "=v"(r.x[0]), "=v"(r.x[1]) : "v"(r.k[0]), "v"(r.c));
        asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %2" : "=v"(r.x[2]), "=v"(r.x[3]) : "v"(r.k[1]));
    } else {
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(r.x[4]), "+v"(r.x[5]) : "v"(r.k[2]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(r.k[3]), "v"(r.k[0]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r.x[6]) : "v"(r.k[1]), "v"(r.k[2]));
        if (odd) asm volatile("v_mov_b32 %0, %1" : "=v"(r.x[14]) : "v"(r.c));
    }
}

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(float* out, int tiles, unsigned long long* ticks, unsigned seed) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    h8 a, b;
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u;
    for (int j = 0; j < 8; j++) {
        s = s * 1664525u + 1013904223u; a[j] = (_Float16)(((int)(s >> 16) & 2047) / 512.0f - 2.0f);
        s = s * 1664525u + 1013904223u; b[j] = (_Float16)(((int)(s >> 16) & 2047) / 512.0f - 2.0f);
    }
    Regs r;
    for (int j = 0; j < 16; j++) r.x[j] = 0.5f + 0.001f * lane + j;
    for (int j = 0; j < 4; j++) { r.k[j] = -0.5f - 0.01f * lane - j; asm volatile("" : "+v"(r.k[j])); }
    r.c = 0.999f; r.y = 0.25f + lane; r.y2 = 0.f; r.s4 = f4{0, 0, 0, 0}; r.s2 = f2{0, 0};
    const unsigned a128 = lane * 16 + (wave & 3) * 1024, a64 = lane * 8 + (wave & 3) * 1024, aw = 65536 + lane * 16 + (wave & 3) * 1024;
    asm volatile("" : "+v"(r.c), "+v"(r.y));
    const bool roleB = MODE == 2 && wave >= 4;
    const unsigned long long t0 = wall_clock64();
    for (int t = 0; t < tiles; t++) {
        if (MODE <= 1 || MODE == 3) {          // 64 MFMAs per tile and wave, the whole mix
#pragma unroll
            for (int g = 0; g < 64; g++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
                if ((g & 1) == 0) { valu_pair(0, r); if ((g & 3) == 0 || MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(r.s4) : "v"(a128));
                                    if (MODE == 3) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.s2) : "v"(a64)); }
                else { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.s2) : "v"(a64)); valu_pair(1, r, (g >> 1) & 1); asm volatile("s_nop 0"); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (!roleB) {                   // role A: 32 S^T MFMAs per tile + the tile's whole VALU work
#pragma unroll
            for (int g = 0; g < 32; g++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
                valu_pair(0, r);
                if ((g & 1) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r.s4) : "v"(a128));
                valu_pair(1, r, g & 1);
                if ((g & 3) == 3) asm volatile("ds_write_b128 %0, %1" : : "v"(aw), "v"(r.s4));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                               // role B: 32 P.V MFMAs per tile, fragment reads, little else
#pragma unroll
            for (int g = 0; g < 32; g++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.s2) : "v"(a64));
                if ((g & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r.s4) : "v"(aw));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        if (MODE == 2) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = wall_clock64();
    float sum = r.s4[0] + r.s4[1] + r.s4[2] + r.s4[3] + r.s2[0] + r.s2[1] + r.y;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) sum += acc[i][j];
    for (int j = 0; j < 16; j++) sum += r.x[j];
    sum += r.y2;
    if (sum == 12345.678f) out[0] = sum + lds[lane];
    if (lane == 0) atomicMax(ticks, t1 - t0);
}

template <int MODE> double run(int grid, int tiles, float* out, unsigned long long* ticks) {
    constexpr int THREADS = MODE == 0 ? 256 : 512;
    auto kfn = probe<MODE, THREADS>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), 128 << 10, 0, out, tiles / 4, ticks, 1u);
    hipDeviceSynchronize();
    hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), 128 << 10, 0, out, tiles, ticks, 2u);
    hipDeviceSynchronize();
    unsigned long long t = 0;
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)tiles * (MODE == 0 ? 64 : (MODE == 1 || MODE == 3) ? 128 : 64);      // per SIMD: mode 1 = two waves x 64, mode 2 = 32 + 32
    return (double)t * 10.0 / mfma_per_simd;
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 2000;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 1024); hipMalloc(&ticks, 64);
    printf("# ns per v_mfma_f32_32x32x16_f16 per SIMD, pseudo-random operands; the tile step's mix per 2 MFMAs: 2 fma, 2 exp, 2 add, 1 cvt_pk, 1 max3, 1/2 mov, 1/2 ds_read_b128, 1 ds_read_b64_tr\n");
    for (int rep = 0; rep < 2; rep++)
        for (int grid : {8, 256}) {
            const double m0 = run<0>(grid, tiles, out, ticks), m1 = run<1>(grid, tiles, out, ticks), m3 = run<3>(grid, tiles, out, ticks), m2 = run<2>(grid, tiles, out, ticks);
            printf("grid %3d | one wave per SIMD, whole mix %6.2f | two waves, whole mix each %6.2f (x%.3f) | ... with a 32-row wave's LDS reads (twice per MFMA) %6.2f (x%.3f) | two waves by ROLE %6.2f (x%.3f)\n",
                   grid, m0, m1, m0 / m1, m3, m0 / m3, m2, m0 / m2);
            fflush(stdout);
        }
    return 0;
}
