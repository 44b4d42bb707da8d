// Skeleton of an anti-phased 8-wave attention main loop on gfx950: what does the STRUCTURE allow, before any real data
// movement is attached?  512 threads = 8 waves = two halves (waves 0-3 / 4-7: one wave of each half per SIMD).
// Every wave alternates
//     load cluster    : NL x ds_read_b128 (K or V^T fragments -> registers), wait
//     compute cluster : 16 x v_mfma_f32_32x32x16_f16, each followed by NV VALU ops (NE of every 4 are v_exp_f32)
// with one s_barrier per cluster boundary; the halves are offset by one cluster, so on every SIMD one wave computes while
// its partner loads.  Variants: ANTI=0 puts both halves in the same phase (what a single per-tile barrier does),
// PRIO=1 raises the second half's priority.
//   hipcc --offload-arch=gfx950 -O3 tools/attn_skeleton_probe.cpp -o attn_skeleton_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NE, int NL, int ANTI, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) ((float*)lds)[i] = seed * i;
    __syncthreads();
    f16x8 b;
    for (int j = 0; j < 8; j++) b[j] = (_Float16)(0.002f * (lane - j));
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float x[16];
    for (int j = 0; j < 16; j++) x[j] = seed + 0.01f * j;
    const float c1 = 0.999f, c2 = 0.001f;
    f32x4 frag[16];
    for (int j = 0; j < 16; j++) frag[j] = (f32x4){seed, seed, seed, seed};
    // conflict-free ds_read_b128 addresses (XOR swizzle like a K tile)
    const unsigned base = (unsigned)(size_t)lds + (lane & 31) * 256 + ((((lane >> 5)) ^ (lane & 15)) << 4);
    const int half = (wave >> 2) & 1;
    if (PRIO && half) __builtin_amdgcn_s_setprio(1);

    auto load_cluster = [&](int it) {
        asm volatile("" ::: "memory");
        const char* src = lds + ((base - (unsigned)(size_t)lds) ^ ((it & 3) << 14));
#pragma unroll
        for (int j = 0; j < NL; j++) frag[j & 15] = *(const f32x4*)(src + ((j & 7) * 32 * 16 + (j >> 3) * 32) % 16384);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("" : "+v"(frag[j]));
    };
    auto compute_cluster = [&]() {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const f16x8 a = __builtin_bit_cast(f16x8, frag[m]);
            if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            else if ((m & 3) == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            else if ((m & 3) == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < NV; n++) {
                const int j = (m * NV + n) & 15;
                if ((n & 3) < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
            }
        }
    };
    if (ANTI) {
        if (half == 0) {
            for (int i = 0; i < iters; i++) {
                load_cluster(i);
                __builtin_amdgcn_s_barrier();
                compute_cluster();
                __builtin_amdgcn_s_barrier();
            }
        } else {
            load_cluster(0);
            for (int i = 0; i < iters; i++) {
                compute_cluster();
                __builtin_amdgcn_s_barrier();
                load_cluster(i + 1);
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        for (int i = 0; i < iters; i++) {
            load_cluster(i);
            compute_cluster();
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    for (int j = 0; j < 16; j++) s += x[j] + frag[j][0];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NV, int NE, int NL, int ANTI, int PRIO> void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, NE, NL, ANTI, PRIO><<<256, 512>>>(out, 50, 0.25f);
    (void)hipEventRecord(e0);
    k<NV, NE, NL, ANTI, PRIO><<<256, 512>>>(out, iters, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 16.0 * 32768.0;
    // cycles per (load+compute) cluster pair per wave at a nominal 2.4 GHz
    printf("%-34s NV=%d (exp %d/4) NL=%2d : %7.3f ms  %7.1f TF  %6.0f ns per cluster pair\n", name, NV, NE, NL, ms, flops / ms / 1e9, ms * 1e6 / iters);
    (void)hipFree(out);
}

int main() {
    run<0, 0, 16, 1, 0>("anti-phase");
    run<0, 0, 16, 0, 0>("in-phase");
    run<3, 1, 16, 1, 0>("anti-phase");
    run<4, 1, 16, 1, 0>("anti-phase");
    run<4, 1, 16, 1, 1>("anti-phase + prio");
    run<4, 1, 16, 0, 0>("in-phase");
    run<5, 1, 16, 1, 0>("anti-phase");
    run<5, 1, 16, 0, 0>("in-phase");
    run<5, 1, 32, 1, 0>("anti-phase");
    run<6, 1, 16, 1, 0>("anti-phase");
    run<6, 2, 16, 1, 0>("anti-phase");
    run<6, 2, 16, 0, 0>("in-phase");
    run<4, 0, 16, 1, 0>("anti-phase");
    run<4, 0, 16, 0, 0>("in-phase");
    return 0;
}
