// Probe (round 6; bench.py's `roofline.other.power_ceiling` / `.hbm_ceiling`, tools/lab/power_ceiling.py): what does the POWER budget leave of the MFMA peak?  (gfx950)
// prefill32_kernel takes 4 % fewer cycles than prefill64_kernel and 2-3 % more time: the part clocks to its power budget, so the ceiling of a
// chip-filling MFMA kernel on random data is not 2.5 PFLOP/s (= 2.4 GHz x 256 CUs x 4 SIMDs x 32x32x16x2 / 32 cycles) but whatever clock the
// board sustains under that load.  This probe measures that ceiling directly: whole-chip streams of v_mfma_f32_32x32x16_f16, one wave per SIMD,
// sustained for ~1 s each (launches of ~40 ms back to back, the last ones reported):
//   stream 0  MFMAs only, four accumulators, every MFMA takes another A / B fragment (8 + 8 fragments in rotation: operand lines toggle as in a GEMM)
//   stream 1  ... plus the prefill tile step's VALU mix per 2 MFMAs (2 fma, 2 exp, 2 add, 1 cvt_pk, 1 max3) — no LDS, no memory
//   stream 2  ... plus its LDS fragment reads (1/2 ds_read_b128 + 1 ds_read_b64_tr_b16 per 2 MFMAs)
// each with pseudo-random fragments and with zero fragments.  Output: TFLOP/s, ns per MFMA per SIMD, and the clock a 32-cycle MFMA implies for
// stream 0 (streams 1 / 2 are issue-bound: their cycles per MFMA are not known a priori).
// Also (the decode kernel's ceiling, same idea): a read-only HBM stream — every workgroup sums a contiguous slice of a 4 GiB buffer with 16-byte
// loads, four in flight per lane, 1 024 x 1 024 threads (the best shape of tools/hbm_read_probe.cpp's sweep, profiles/r01_hbm_read_probe.txt).
// `--quick [seconds]`: what bench.py reports (MFMAs only; MFMAs + VALU mix + LDS reads; the HBM read stream), one JSON object on stdout.
// build: vattention_amd/build.py build_probe() = hipcc --offload-arch=gfx950 -O3 -o tools/power_ceiling_probe tools/power_ceiling_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// REUSE (stream 0 only; `--operand-reuse`): which operand lines toggle between consecutive MFMAs — 0: A and B both change every MFMA;
// 1: A stays for two MFMAs, B alternates between two fragments (the kernel's phase A: one K fragment against the two Q^T blocks; phase B alike);
// 2: B stays for four MFMAs, A changes every MFMA; 3: A and B both stay for four MFMAs (only the accumulator changes)
template <int STREAM, int REUSE = 0>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, unsigned seed, int zero) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    h8 a[8], b[8];
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u;
    for (int f = 0; f < 8; f++)
        for (int j = 0; j < 8; j++) {
            s = s * 1664525u + 1013904223u; a[f][j] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 16) & 2047) / 512.0f - 2.0f);
            s = s * 1664525u + 1013904223u; b[f][j] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 16) & 2047) / 512.0f - 2.0f);
        }
    for (int f = 0; f < 8; f++) asm volatile("" : "+v"(a[f]), "+v"(b[f]));
    float x[8], k[4], c = 0.999f, y = 0.25f + lane;
    for (int j = 0; j < 8; j++) x[j] = zero ? 0.f : 0.5f + 0.001f * lane + j;
    for (int j = 0; j < 4; j++) { k[j] = zero ? 0.f : -0.5f - 0.01f * lane - j; asm volatile("" : "+v"(k[j])); }
    f4 s4 = {0, 0, 0, 0};
    f2 s2 = {0, 0};
    const unsigned a128 = lane * 16 + wave * 1024, a64 = lane * 8 + wave * 1024;
    if (STREAM == 2) {
        for (int i = threadIdx.x; i < 4096; i += 256) ((unsigned*)lds)[i] = zero ? 0u : (s = s * 1664525u + 1013904223u);
        __syncthreads();
    }
    asm volatile("" : "+v"(c), "+v"(y));
    for (int t = 0; t < iters; t++) {
#pragma unroll
        for (int g = 0; g < 64; g++) {
            // the accumulators stay small: B alternates sign through the rotation (fragments are +-2), sums random-walk
            if (REUSE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a[g & 7]), "v"(b[(g * 5 + (g >> 3)) & 7]));
            if (REUSE == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a[(g >> 1) & 7]), "v"(b[(g & 1) + 2 * ((g >> 4) & 3)]));
            if (REUSE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a[g & 7]), "v"(b[(g >> 2) & 7]));
            if (REUSE == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a[(g >> 2) & 7]), "v"(b[((g >> 2) * 3) & 7]));
            if (STREAM >= 1) {
                if ((g & 1) == 0) {
                    asm volatile("v_fma_f32 %0, %2, %3, %3\n\tv_fma_f32 %1, %2, %3, %3" : "=v"(x[0]), "=v"(x[1]) : "v"(k[0]), "v"(c));
                    asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %2" : "=v"(x[2]), "=v"(x[3]) : "v"(k[1]));
                    if (STREAM == 2 && (g & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(s4) : "v"(a128));
                } else {
                    if (STREAM == 2) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(s2) : "v"(a64));
                    asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(x[4]), "+v"(x[5]) : "v"(k[2]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y) : "v"(k[3]), "v"(k[0]));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[6]) : "v"(k[1]), "v"(k[2]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STREAM == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float sum = s4[0] + s4[1] + s4[2] + s4[3] + s2[0] + s2[1] + y;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) sum += acc[i][j];
    for (int j = 0; j < 8; j++) sum += x[j];
    if (sum == 12345.678f) out[0] = sum + lds[lane];
}

__global__ __launch_bounds__(1024) void hbm_read(const uint4* __restrict__ src, float* out, size_t n16) {
    const size_t per = n16 / gridDim.x;
    const uint4* p = src + (size_t)blockIdx.x * per;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + 3 * (size_t)blockDim.x < per; i += 4 * (size_t)blockDim.x) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}
// GB/s of the read stream: mean of 20 back-to-back passes over 4 GiB (16x the 256 MiB MALL), after two warm-up passes
static double hbm_read_stream(float* out) {
    const size_t bytes = 4ull << 30;
    uint4* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return 0.0;
    (void)hipMemset(d, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(hbm_read, dim3(1024), dim3(1024), 0, 0, d, out, bytes / 16);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(hbm_read, dim3(1024), dim3(1024), 0, 0, d, out, bytes / 16);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return 20.0 * bytes / ms / 1e6;
}

static double g_seconds = 1.0;
static bool g_quiet = false;
template <int STREAM, int REUSE = 0> double run(const char* what, int zero, float* out) {
    auto kfn = probe<STREAM, REUSE>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
    const int grid = 256, iters = 20000;      // 64 x 20 000 MFMAs per wave: ~40-60 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double last[3] = {0, 0, 0};
    const auto w0 = std::chrono::steady_clock::now();
    const double epoch0 = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < g_seconds || n < 3) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 64 << 10, 0, out, iters, 1u + n, zero);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        last[n % 3] = ms;
        n++;
    }
    const double ms = (last[0] + last[1] + last[2]) / 3.0;
    const double mfma = (double)grid * 4 * 64.0 * iters;                      // MFMAs of the launch
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    const double ns = ms * 1e6 / (64.0 * iters);
    if (g_quiet) return tf;
    printf("%-34s %-7s %8.1f TFLOP/s  (%.3f of 2500)  %6.2f ns per MFMA per SIMD", what, zero ? "zeros" : "random", tf, tf / 2500.0, ns);
    if (STREAM == 0) printf("  => %.0f MHz at 32 cycles per MFMA", 32.0 / ns * 1e3);
    const double epoch1 = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    printf("   [%d launches, last three %.2f %.2f %.2f ms] t0=%.3f t1=%.3f\n", n, last[0], last[1], last[2], epoch0 + 0.5 * (epoch1 - epoch0), epoch1);
    fflush(stdout);
    return tf;
}

int main(int argc, char** argv) {
    float* out;
    if (hipMalloc(&out, 1024) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    if (argc > 1 && !strcmp(argv[1], "--quick")) {
        g_seconds = argc > 2 ? atof(argv[2]) : 0.7;
        g_quiet = true;
        const double a = run<0>("", 0, out), b = run<2>("", 0, out);
        const double h = hbm_read_stream(out);
        printf("{\"mfma_only_tflops\": %.1f, \"tile_step_stream_tflops\": %.1f, \"seconds_each\": %.2f, \"hbm_read_stream_gbs\": %.1f}\n", a, b, g_seconds, h);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--operand-reuse")) {
        printf("# MFMAs only, whole chip, random operands: which operand lines toggle between consecutive MFMAs\n");
        for (int rep = 0; rep < 2; rep++) {
            run<0, 0>("A and B change every MFMA", 0, out);
            run<0, 1>("A stays for 2, B alternates", 0, out);
            run<0, 2>("B stays for 4, A changes", 0, out);
            run<0, 3>("A and B stay for 4", 0, out);
        }
        return 0;
    }
    printf("# whole chip (256 workgroups x 4 waves, one wave per SIMD), v_mfma_f32_32x32x16_f16, ~1 s sustained per line\n");
    for (int rep = 0; rep < 2; rep++) {
        run<0>("MFMAs only", 0, out);
        run<1>("MFMAs + the tile step's VALU mix", 0, out);
        run<2>("MFMAs + VALU mix + LDS reads", 0, out);
        run<0>("MFMAs only", 1, out);
        run<1>("MFMAs + the tile step's VALU mix", 1, out);
    }
    for (int rep = 0; rep < 2; rep++) {
        const double h = hbm_read_stream(out);
        printf("HBM read stream (4 GiB, 1024 x 1024 threads, 4 x 16 B in flight per lane)  %8.1f GB/s  (%.3f of 8000)\n", h, h / 8000.0);
    }
    return 0;
}
