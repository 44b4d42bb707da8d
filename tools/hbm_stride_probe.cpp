// Does the [token][kv head][d] layout cost HBM efficiency?  Each workgroup streams one "head": the h-th 256-byte piece of every
// row of ROWB bytes (ROWB = 256 * heads), against workgroups that stream the same number of bytes contiguously.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_stride_probe.cpp -o hbm_stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
// grid: (splits, heads); block 256 = 4 waves; wave w of split s reads tiles s*4+w, +4*splits ... of 32 rows each;
// lane (l15 = lane&15 -> 16-byte chunk of the 256-byte piece, r = lane>>4 -> row within a group of 4)
template <int HEADS, bool CONTIG>
__global__ __launch_bounds__(256, 3) void rd(const uint4* __restrict__ src, float* out, size_t rows, int splits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, s = blockIdx.x;
    const size_t row16 = HEADS * 16;                       // uint4 per row
    unsigned acc = 0;
    const size_t tiles = rows / 32;
    if (CONTIG) {
        // same bytes per workgroup, but one contiguous slice: (head h, split s) -> slice index h*splits + s
        const size_t per = rows * 16 / splits;             // uint4 per workgroup = (rows/splits) * 256 B / 16
        const uint4* p = src + ((size_t)h * splits + s) * per;
        for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
            uint4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
            acc += a.x ^ b.x ^ c.x ^ d.x;
        }
    } else {
        const size_t per_split = (tiles + splits - 1) / splits;
        const size_t t0 = (size_t)s * per_split, t1 = t0 + per_split < tiles ? t0 + per_split : tiles;
        for (size_t t = t0 + wave; t < t1; t += 4) {
            const uint4* p = src + (t * 32) * row16 + (size_t)h * 16 + (lane & 15);
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = p[(size_t)((lane >> 4) + 4 * j) * row16];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].w;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}
template <int HEADS, bool CONTIG> void run(const uint4* d, float* out, size_t bytes, int splits) {
    const size_t rows = bytes / (256 * HEADS);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd<HEADS, CONTIG><<<dim3(splits, HEADS), 256>>>(d, out, rows, splits);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) rd<HEADS, CONTIG><<<dim3(splits, HEADS), 256>>>(d, out, rows, splits);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("heads %d  %-22s splits %4d (%5d workgroups): %7.1f GB/s\n", HEADS, CONTIG ? "contiguous slices" : "one head per workgroup", splits, splits * HEADS, 5.0 * bytes / ms / 1e6);
}
int main() {
    const size_t bytes = 2ull << 30;
    uint4* d; float* out;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(d, 1, bytes);
    for (int splits : {192, 768}) {
        run<4, false>(d, out, bytes, splits); run<4, true>(d, out, bytes, splits);
    }
    for (int splits : {96, 384}) { run<8, false>(d, out, bytes, splits); run<8, true>(d, out, bytes, splits); }
    for (int splits : {768, 3072}) { run<1, false>(d, out, bytes, splits); run<1, true>(d, out, bytes, splits); }
    return 0;
}
