// Read-only HBM streaming ceiling on MI355X: every workgroup sums a contiguous slice with 16-byte loads (several in flight
// per lane), nothing is written back but one float per workgroup.  Sweeps workgroup count / size and unroll depth.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_probe.cpp -o hbm_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNROLL>
__global__ void rd(const uint4* __restrict__ src, float* out, size_t n16) {
    const size_t per = n16 / gridDim.x;
    const uint4* p = src + (size_t)blockIdx.x * per;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * blockDim.x < per; i += (size_t)UNROLL * blockDim.x) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}
template <int UNROLL> void run(const uint4* d, float* out, size_t bytes, int grid, int block) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd<UNROLL><<<grid, block>>>(d, out, bytes / 16);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) rd<UNROLL><<<grid, block>>>(d, out, bytes / 16);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid %5d x %4d  unroll %d : %7.1f GB/s\n", grid, block, UNROLL, 5.0 * bytes / ms / 1e6);
}
int main() {
    const size_t bytes = 4ull << 30;
    uint4* d; float* out;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(d, 1, bytes);
    for (int grid : {256, 512, 768, 1024, 2048, 4096, 16384})
        for (int block : {256, 512, 1024}) { run<4>(d, out, bytes, grid, block); }
    run<8>(d, out, bytes, 768, 256); run<8>(d, out, bytes, 2048, 256); run<8>(d, out, bytes, 1024, 512);
    run<2>(d, out, bytes, 2048, 256); run<2>(d, out, bytes, 4096, 512); run<1>(d, out, bytes, 16384, 256);
    return 0;
}
