// MFMA issue-rate probe: v_mfma_f32_32x32x16_f16 with 1/2/4/8 independent accumulator chains, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH> __global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
    f32x16 acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16 / CH; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> void run(int threads, const char* name) {
    float* out; hipMalloc(&out, 256 * 4 * 512 * 4);
    const int iters = 20000, blocks = 256 * (512 / threads) ;   // one 512-thread block (2 waves/SIMD) or two... keep 1 block per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH><<<256, threads>>>(out, 100);
    hipEventRecord(e0);
    k<CH><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 16.0 * 32768.0;
    printf("%-10s chains %d, %d waves/CU: %.1f TFLOP/s\n", name, CH, threads / 64, flops / ms / 1e9);
    hipFree(out); (void)blocks;
}
int main() {
    run<1>(256, "1w/SIMD"); run<2>(256, "1w/SIMD"); run<4>(256, "1w/SIMD"); run<8>(256, "1w/SIMD");
    run<1>(512, "2w/SIMD"); run<2>(512, "2w/SIMD"); run<4>(512, "2w/SIMD"); run<8>(512, "2w/SIMD");
    return 0;
}
