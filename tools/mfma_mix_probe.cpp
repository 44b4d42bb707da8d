// How many independent VALU / transcendental / LDS-read instructions fit next to each v_mfma_f32_32x32x16_f16
// before the MFMA rate drops?  8 waves per CU (2 per SIMD), 4 accumulate chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND> __global__ __launch_bounds__(512, 2) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
    f32x16 acc[4];
    for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float v[16];
    for (int j = 0; j < 16; j++) v[j] = seed + j;
    lds[threadIdx.x] = seed; __syncthreads();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; n++) {
                const int j = (c * NV + n) & 15;
                if (KIND == 0) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
                else if (KIND == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);
                else { v[j] += lds[(threadIdx.x * 4 + j * 64 + i) & 4095]; }
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    for (int j = 0; j < 16; j++) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int KIND> void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, KIND><<<256, 512>>>(out, 100, 0.25f);
    hipEventRecord(e0);
    k<NV, KIND><<<256, 512>>>(out, iters, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 4.0 * 32768.0;
    printf("%-8s x%2d per MFMA: %.1f TFLOP/s\n", name, NV, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0, 0>("v_fma"); run<2, 0>("v_fma"); run<4, 0>("v_fma"); run<6, 0>("v_fma"); run<8, 0>("v_fma"); run<12, 0>("v_fma"); run<16, 0>("v_fma");
    run<1, 1>("v_exp"); run<2, 1>("v_exp"); run<4, 1>("v_exp"); run<8, 1>("v_exp");
    run<1, 2>("ds_read"); run<2, 2>("ds_read"); run<4, 2>("ds_read");
    return 0;
}
