// issue_probe2 — does a SECOND wave per SIMD recover the matrix-pipe time that one wave loses to instruction issue?  (gfx950)
#pragma clang diagnostic ignored "-Wunused-value"
// prefill64_kernel runs one 512-register wave per SIMD and is bound by what that single wave can issue beside its MFMAs
// (profiles/r02_issue_probe.txt).  This probe times the SAME per-tile instruction mix (per v_mfma_f32_32x32x16_f16: 1 v_fma, 1 v_exp,
// 1 v_add, 1/2 v_cvt_pk, 1/2 v_max3, 1/4 v_mov, 1/4 ds_read_b128, 1/2 ds_read_b64_tr_b16, 1 s_nop) in four arrangements:
//   mode 0  one wave per SIMD, the mix interleaved between the MFMAs (what prefill64 does today)
//   mode 1  two waves per SIMD (512-thread workgroup), each wave the interleaved stream
//   mode 2  two waves per SIMD, each wave alternating a block of NB MFMAs with the block's VALU/LDS work (no intra-wave overlap), no stagger
//   mode 3  as 2, the second wave of a SIMD starts half a period late
//   mode 4  as 3, s_setprio 1 around the MFMA block
// Output: ns per MFMA per SIMD; the floor is the MFMA-only figure (13.6 ns at boost clocks, about 18 ns power-limited on the whole chip).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe2 tools/issue_probe2.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Regs {
    float x[8];
    float y, c;
    f4 s4;
    f2 s2;
};

// the non-MFMA work that belongs to FOUR MFMAs, slice q (0..3) of it
template <bool LDS>
__device__ __forceinline__ void mix_slice(int q, Regs& r, unsigned a128, unsigned a64) {
    if (q == 0) {
        asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(r.x[0]), "+v"(r.x[1]) : "v"(r.c));
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(r.x[2]), "+v"(r.x[3]));
        if (LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(r.s4) : "v"(a128));
        asm volatile("s_nop 0");
    } else if (q == 1) {
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(r.x[4]), "+v"(r.x[5]) : "v"(r.c));
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(r.x[6]), "+v"(r.x[7]));
        if (LDS) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.s2) : "v"(a64));
        asm volatile("s_nop 0");
    } else if (q == 2) {
        asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(r.x[2]), "+v"(r.x[3]) : "v"(r.c));
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(r.x[0]), "+v"(r.x[1]) : "v"(r.c));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(r.x[4]), "v"(r.x[5]));
        if (LDS) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.s2) : "v"(a64));
        asm volatile("s_nop 0");
    } else {
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(r.x[6]), "v"(r.x[7]));
        asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %2, %3" : "+v"(r.x[4]), "+v"(r.x[5]) : "v"(r.c), "v"(r.x[0]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(r.x[6]) : "v"(r.c));
        asm volatile("s_nop 0");
    }
}

template <int MODE, int NB, bool LDS, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(float* out, int iters, unsigned long long* ticks) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    Regs r;
    for (int j = 0; j < 8; j++) r.x[j] = 0.5f + 0.001f * lane + j;
    r.c = 0.999f;
    r.y = 0.25f + lane;
    r.s4 = f4{0, 0, 0, 0};
    r.s2 = f2{0, 0};
    const unsigned a128 = lane * 16 + wave * 1024, a64 = lane * 8 + wave * 1024;
    asm volatile("" : "+v"(r.c), "+v"(r.y));
    if (MODE >= 3 && wave >= 4) {          // half a period of VALU work first
#pragma unroll
        for (int g = 0; g < NB; g++) mix_slice<false>(g & 3, r, a128, a64);
    }
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE <= 1) {
#pragma unroll
            for (int g = 0; g < NB; g++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
                mix_slice<LDS>(g & 3, r, a128, a64);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (MODE == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < NB; g++) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
            if (MODE == 4) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NB; g++) mix_slice<LDS>(g & 3, r, a128, a64);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = wall_clock64();
    float s = r.s4[0] + r.s4[1] + r.s4[2] + r.s4[3] + r.s2[0] + r.s2[1] + r.y;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int j = 0; j < 8; j++) s += r.x[j];
    if (s == 12345.678f) out[0] = s + lds[lane];
    if (lane == 0) atomicMax(ticks, t1 - t0);      // the slowest wave of the launch (the two waves of a SIMD need not finish together)
}

template <int MODE, int NB, bool LDS>
double run(int grid, int iters, float* out, unsigned long long* ticks) {
    constexpr int THREADS = MODE == 0 ? 256 : 512;
    auto kfn = probe<MODE, NB, LDS, THREADS>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), 96 << 10, 0, out, iters / 4, ticks);
    hipDeviceSynchronize();
    hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), 96 << 10, 0, out, iters, ticks);
    hipDeviceSynchronize();
    unsigned long long t = 0;
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * NB * (MODE == 0 ? 1 : 2);
    return (double)t * 10.0 / mfma_per_simd;      // wall_clock64: 100 MHz -> ns per MFMA per SIMD
}

template <int NB, bool LDS>
void row(int grid, int iters, float* out, unsigned long long* ticks) {
    const double m0 = run<0, NB, LDS>(grid, iters, out, ticks);
    const double m1 = run<1, NB, LDS>(grid, iters, out, ticks);
    const double m2 = run<2, NB, LDS>(grid, iters, out, ticks);
    const double m3 = run<3, NB, LDS>(grid, iters, out, ticks);
    const double m4 = run<4, NB, LDS>(grid, iters, out, ticks);
    printf("grid %3d  block of %2d MFMAs, LDS reads %s | ns per MFMA per SIMD: 1 wave interleaved %6.2f | 2 waves interleaved %6.2f | 2 waves serial blocks %6.2f, staggered %6.2f, staggered + setprio %6.2f\n",
           grid, NB, LDS ? "yes" : "no ", m0, m1, m2, m3, m4);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1024);
    hipMalloc(&ticks, 64);
    printf("# per v_mfma_f32_32x32x16_f16: 1 v_fma, 1 v_exp, 1 v_add, 1/2 v_cvt_pk, 1/2 v_max3, 1/4 v_mov, 1 s_nop (+ 1/4 ds_read_b128, 1/2 ds_read_b64_tr_b16)\n");
    for (int grid : {8, 256}) {
        row<16, false>(grid, iters, out, ticks);
        row<32, false>(grid, iters, out, ticks);
        row<16, true>(grid, iters, out, ticks);
        row<32, true>(grid, iters, out, ticks);
        row<64, true>(grid, iters, out, ticks);
    }
    return 0;
}
