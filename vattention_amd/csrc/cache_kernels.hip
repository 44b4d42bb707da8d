// cache_flat / append: contiguous KV append (16-byte vector copies).  gfx950.
#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// cache_flat / append
// ============================================================================================

// One 16-byte chunk per thread; K and V rows copied by the same launch (cache_kernels.cu:483-520).
__global__ void cache_flat_vec_kernel(const uint4* __restrict__ key, const uint4* __restrict__ value,
                                      uint4* __restrict__ k_cache, uint4* __restrict__ v_cache,
                                      int64_t num_tokens, int chunks_per_row, int64_t key_stride, int64_t value_stride,
                                      int64_t k_cache_stride, int64_t v_cache_stride) {
    // strides are in 16-byte chunks here
    const int64_t total = num_tokens * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / chunks_per_row;
        const int c = (int)(i - t * chunks_per_row);
        const uint4 kv = key[t * key_stride + c];
        const uint4 vv = value[t * value_stride + c];
        k_cache[t * k_cache_stride + c] = kv;
        v_cache[t * v_cache_stride + c] = vv;
    }
}

template <typename E>
__global__ void cache_flat_scalar_kernel(const E* __restrict__ key, const E* __restrict__ value, E* __restrict__ k_cache,
                                         E* __restrict__ v_cache, int64_t num_tokens, int n, int64_t key_stride,
                                         int64_t value_stride, int64_t k_cache_stride, int64_t v_cache_stride) {
    const int64_t total = num_tokens * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / n;
        const int c = (int)(i - t * n);
        k_cache[t * k_cache_stride + c] = key[t * key_stride + c];
        v_cache[t * v_cache_stride + c] = value[t * value_stride + c];
    }
}

// Append of k_new/v_new [b, sn, h_k, d] at row cache_seqlens[b] of slot cache_batch_idx[b]
// (flash_attn_interface.py:1168-1176).  16-byte chunks; d*itemsize is a multiple of 16.
__global__ void append_kv_kernel(vattn_attn_params p) {
    const int b = blockIdx.y;
    const int slot = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
    const int len = p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k;
    const int cpr = p.d / 8;                       // 16-byte chunks per head row
    const int total = p.seqlen_knew * p.h_k * cpr;
    const uint16_t* kn = (const uint16_t*)p.k_new;
    const uint16_t* vn = (const uint16_t*)p.v_new;
    uint16_t* kc = (uint16_t*)p.k_cache;
    uint16_t* vc = (uint16_t*)p.v_cache;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % cpr;
        const int hk = (i / cpr) % p.h_k;
        const int t = i / (cpr * p.h_k);
        const int row = len + t;
        if (row >= p.seqlen_k) continue;           // never write past the cache view
        const uint4 kv = *(const uint4*)(kn + b * p.knew_batch_stride + t * p.knew_row_stride + hk * p.knew_head_stride + c * 8);
        const uint4 vv = *(const uint4*)(vn + b * p.vnew_batch_stride + t * p.vnew_row_stride + hk * p.vnew_head_stride + c * 8);
        *(uint4*)(kc + (int64_t)slot * p.k_batch_stride + (int64_t)row * p.k_row_stride + hk * p.k_head_stride + c * 8) = kv;
        *(uint4*)(vc + (int64_t)slot * p.v_batch_stride + (int64_t)row * p.v_row_stride + hk * p.v_head_stride + c * 8) = vv;
    }
}

void launch_append(const vattn_attn_params* p, hipStream_t st) {
    const int total = p->seqlen_knew * p->h_k * (p->d / 8);
    dim3 grid((total + 255) / 256, p->b), block(256);
    hipLaunchKernelGGL(append_kv_kernel, grid, block, 0, st, *p);
}

}  // namespace vattn_k

using namespace vattn_k;

extern "C" {

int vattn_cache_flat(const void* key, const void* value, void* k_cache, void* v_cache, int64_t num_tokens,
                     int32_t num_heads, int32_t head_size, int64_t key_stride, int64_t value_stride,
                     int64_t k_cache_stride, int64_t v_cache_stride, int32_t itemsize, void* stream) {
    if (num_tokens <= 0) return VATTN_K_OK;
    if (!key || !value || !k_cache || !v_cache) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)num_heads * head_size;
    const int64_t row_bytes = n * itemsize;
    const bool vec = row_bytes % 16 == 0 && (key_stride * itemsize) % 16 == 0 && (value_stride * itemsize) % 16 == 0 &&
                     (k_cache_stride * itemsize) % 16 == 0 && (v_cache_stride * itemsize) % 16 == 0 &&
                     ((((uintptr_t)key) | ((uintptr_t)value) | ((uintptr_t)k_cache) | ((uintptr_t)v_cache)) & 15) == 0;
    if (vec) {
        const int cpr = (int)(row_bytes / 16);
        const int64_t total = num_tokens * cpr;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        const int64_t f = 16 / itemsize;
        hipLaunchKernelGGL(cache_flat_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)key, (const uint4*)value,
                           (uint4*)k_cache, (uint4*)v_cache, num_tokens, cpr, key_stride / f, value_stride / f,
                           k_cache_stride / f, v_cache_stride / f);
    } else {
        const int64_t total = num_tokens * n;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (itemsize == 2)
            hipLaunchKernelGGL(cache_flat_scalar_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint16_t*)key,
                               (const uint16_t*)value, (uint16_t*)k_cache, (uint16_t*)v_cache, num_tokens, (int)n, key_stride,
                               value_stride, k_cache_stride, v_cache_stride);
        else if (itemsize == 4)
            hipLaunchKernelGGL(cache_flat_scalar_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t*)key,
                               (const uint32_t*)value, (uint32_t*)k_cache, (uint32_t*)v_cache, num_tokens, (int)n, key_stride,
                               value_stride, k_cache_stride, v_cache_stride);
        else
            return fail(VATTN_K_ERR_UNSUPPORTED, "cache_flat supports 2- and 4-byte element types");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

}  // extern "C"
