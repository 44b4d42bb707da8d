// Matrix-pipe ceiling as a function of OPERAND DATA on gfx950.  Back-to-back v_mfma_f32_32x32x16_f16, 2 waves per SIMD,
// 4 accumulate chains; the A/B fragments rotate through 8 register sets per lane so that consecutive instructions see
// different operands.  Data: zeros / one constant / N(0,1) fp16 / N(0,1) scaled by 1/64.  The board is power-managed:
// identical instruction streams run at different clocks depending on how many bits toggle.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.cpp -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void k(const f16x8* __restrict__ src, float* out, int iters) {
    f16x8 a[8], b[8];
    for (int j = 0; j < 8; j++) { a[j] = src[(j * 512 + threadIdx.x)]; b[j] = src[((8 + j) * 512 + threadIdx.x)]; }
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a[j]), "v"(b[j]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a[(j + 1) & 7]), "v"(b[(j + 3) & 7]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a[(j + 2) & 7]), "v"(b[(j + 5) & 7]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a[(j + 3) & 7]), "v"(b[(j + 7) & 7]));
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static double gauss() {
    double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

int main() {
    const size_t n = 16 * 512 * 8;
    std::vector<_Float16> h(n);
    _Float16* d; float* out;
    (void)hipMalloc(&d, n * 2); (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"zeros", "constant 0.5", "N(0,1)", "N(0,1)/64", "N(0,1), long run"};
    for (int mode = 0; mode < 5; mode++) {
        srand(1234);
        for (size_t i = 0; i < n; i++) {
            double v = mode == 0 ? 0.0 : mode == 1 ? 0.5 : mode == 3 ? gauss() / 64 : gauss();
            h[i] = (_Float16)v;
        }
        (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        const int iters = mode == 4 ? 40000 : 4000;
        k<<<256, 512>>>((const f16x8*)d, out, 100);
        (void)hipEventRecord(e0);
        k<<<256, 512>>>((const f16x8*)d, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 8 * iters * 32.0 * 32768.0;
        printf("%-18s : %8.3f ms  %7.1f TFLOP/s\n", names[mode], ms, flops / ms / 1e9);
    }
    return 0;
}
