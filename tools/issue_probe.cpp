// issue_probe — what one wave per SIMD can issue next to a stream of v_mfma_f32_32x32x16_f16 on gfx950.
// For every instruction kind X: time {1 MFMA + NV x X} groups (NV = 0, 2, 4, 6, 8) and X alone, one 256-thread workgroup per CU
// (1 wave per SIMD, as the prefill64 kernel runs), on the whole chip (power-limited clocks) and on 8 CUs (boost clocks).
// Output: ns per group and the MFMA rate it leaves.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe tools/issue_probe.cpp
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum Kind { FMA, EXP, ADD, DOT2C_F16, CVT_PK_F16, MAX3, PK_FMA_F32, PK_ADD_F32, ACC_READ, ADD_U32, S_NOP0, DS_B128, DS_TR_B64,
            PK_MUL_F32, EXP_F16, PK_FMA_F16, S_MOV, DOT2_BF16, CVT_PK_BF16, PERM, EXP_ADD_FMA, NKINDS };
static const char* kNames[] = {"v_fma_f32", "v_exp_f32", "v_add_f32", "v_dot2c_f32_f16", "v_cvt_pk_f16_f32", "v_max3_f32", "v_pk_fma_f32",
                               "v_pk_add_f32", "v_accvgpr_read", "v_add_u32", "s_nop 0", "ds_read_b128", "ds_read_b64_tr_b16",
                               "v_pk_mul_f32", "v_exp_f16", "v_pk_fma_f16", "s_mov_b32", "v_dot2_f32_bf16", "v_cvt_pk_bf16_f32",
                               "v_perm_b32", "fma+exp+add mix"};

template <int KIND>
__device__ __forceinline__ void op(float& x, float& y, f2& p, float c, unsigned lds_addr, float (&sink)[4], int j) {
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    else if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if constexpr (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    else if constexpr (KIND == DOT2C_F16) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(x) : "s"(0x3c003c00), "v"(c));
    else if constexpr (KIND == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(c), "v"(y));
    else if constexpr (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(y));
    else if constexpr (KIND == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
    else if constexpr (KIND == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p));
    else if constexpr (KIND == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
    else if constexpr (KIND == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x));
    else if constexpr (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
    else if constexpr (KIND == S_NOP0) asm volatile("s_nop 0");
    else if constexpr (KIND == S_MOV) asm volatile("s_mov_b32 s40, s41" ::: "s40");
    else if constexpr (KIND == DS_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(*(f4*)sink) : "v"(lds_addr));
    else if constexpr (KIND == DS_TR_B64) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(*(f2*)sink) : "v"(lds_addr));
    else if constexpr (KIND == EXP_F16) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
    else if constexpr (KIND == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    else if constexpr (KIND == DOT2_BF16) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(c), "v"(y));
    else if constexpr (KIND == CVT_PK_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(c), "v"(y));
    else if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(y));
    else if constexpr (KIND == EXP_ADD_FMA) {
        if (j % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
        else if (j % 3 == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    }
}

template <int KIND, int NV, bool MFMA>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, unsigned long long* ticks) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    float x[8];
    f2 p[8];
    for (int j = 0; j < 8; j++) { x[j] = 0.5f + 0.001f * lane + j; p[j] = f2{x[j], x[j] * 0.5f}; }
    float c = 0.999f, y = 0.25f + lane;
    float sink[4] = {0, 0, 0, 0};
    const unsigned lds_addr = (KIND == DS_TR_B64 ? lane * 8 : lane * 16) + (threadIdx.x >> 6) * 1024;
    asm volatile("" : "+v"(c), "+v"(y));
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; j++) op<KIND>(x[(g * NV + j) & 7], y, p[(g * NV + j) & 7], c, lds_addr, sink, g * NV + j);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KIND == DS_B128 || KIND == DS_TR_B64) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = wall_clock64();
    float s = sink[0] + sink[1] + sink[2] + sink[3] + y;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int j = 0; j < 8; j++) s += x[j] + p[j][0] + p[j][1];
    if (s == 12345.678f) out[0] = s + lds[lane];
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

struct Res { float ms; double wall_ns; };

template <int KIND, int NV, bool MFMA>
Res run(int grid, int iters, float* out, unsigned long long* ticks) {
    auto kfn = probe<KIND, NV, MFMA>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 96 << 10, 0, out, iters / 4, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 96 << 10, 0, out, iters, ticks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t = 0;
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    return Res{ms, (double)t * 10.0};      // wall_clock64: 100 MHz
}

template <int KIND>
void sweep(int grid, int iters, float* out, unsigned long long* ticks) {
    const double groups = (double)iters * 8;
    Res r0 = run<KIND, 0, true>(grid, iters, out, ticks);
    Res r2 = run<KIND, 2, true>(grid, iters, out, ticks);
    Res r4 = run<KIND, 4, true>(grid, iters, out, ticks);
    Res r6 = run<KIND, 6, true>(grid, iters, out, ticks);
    Res r8 = run<KIND, 8, true>(grid, iters, out, ticks);
    Res rv = run<KIND, 8, false>(grid, iters, out, ticks);
    const double base = r0.wall_ns / groups;
    printf("%-22s grid %3d  ns/group: mfma-only %6.2f  +2 %6.2f  +4 %6.2f  +6 %6.2f  +8 %6.2f | alone ns/inst %5.2f | in MFMA-times: alone %5.3f  slope(4->8) %5.3f\n",
           kNames[KIND], grid, base, r2.wall_ns / groups, r4.wall_ns / groups, r6.wall_ns / groups, r8.wall_ns / groups,
           rv.wall_ns / groups / 8, rv.wall_ns / groups / 8 / base, (r8.wall_ns - r4.wall_ns) / groups / 4 / base);
    fflush(stdout);
}

template <int K>
void all(int grid, int iters, float* out, unsigned long long* ticks) {
    if constexpr (K < NKINDS) {
        sweep<K>(grid, iters, out, ticks);
        all<K + 1>(grid, iters, out, ticks);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1024);
    hipMalloc(&ticks, 64);
    printf("# one 256-thread workgroup per CU (1 wave / SIMD); group = 1 v_mfma_f32_32x32x16_f16 + NV x the instruction\n");
    printf("# a 32-cycle MFMA at 2.4 GHz is 13.3 ns; 'in MFMA-times' = issue cost relative to the measured mfma-only group\n");
    all<0>(8, iters, out, ticks);
    all<0>(256, iters, out, ticks);
    return 0;
}
