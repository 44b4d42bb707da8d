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

// cache_flat with the rotary embedding of the key rows fused in (include/vattn_kernels.h): one thread per (token, head, 8-element
// chunk of the FIRST half): it owns the chunk and its NeoX partner chunk (d + hs/2), rotates both and stores both; the V row
// is copied by the same thread (two chunks).
template <typename T>
__global__ void cache_flat_rope_kernel(const T* __restrict__ key, const T* __restrict__ value, T* __restrict__ k_cache, T* __restrict__ v_cache,
                                       int64_t num_tokens, int num_heads, int hs, int64_t key_stride, int64_t value_stride,
                                       int64_t k_cache_stride, int64_t v_cache_stride, const T* __restrict__ cos_sin, int64_t cs_stride, int64_t pos0) {
    using V8 = typename Tr<T>::v8;
    const int half_chunks = hs / 16;
    const int64_t total = num_tokens * num_heads * half_chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % half_chunks);
        const int h = (int)((i / half_chunks) % num_heads);
        const int64_t t = i / ((int64_t)half_chunks * num_heads);
        const int d0 = 8 * c;
        const T* krow = key + t * key_stride + (int64_t)h * hs;
        const T* vrow = value + t * value_stride + (int64_t)h * hs;
        V8 x = as_v8<V8>(*(const uint4*)(krow + d0)), y = as_v8<V8>(*(const uint4*)(krow + hs / 2 + d0));
        const T* cs = cos_sin + (pos0 + t) * cs_stride;
        const V8 cc = as_v8<V8>(*(const uint4*)(cs + d0)), ss = as_v8<V8>(*(const uint4*)(cs + hs / 2 + d0));
        rope8<T>(x, y, cc, ss);
        T* kdst = k_cache + t * k_cache_stride + (int64_t)h * hs;
        T* vdst = v_cache + t * v_cache_stride + (int64_t)h * hs;
        uint4 ux, uy;
        __builtin_memcpy(&ux, &x, 16);
        __builtin_memcpy(&uy, &y, 16);
        *(uint4*)(kdst + d0) = ux;
        *(uint4*)(kdst + hs / 2 + d0) = uy;
        *(uint4*)(vdst + d0) = *(const uint4*)(vrow + d0);
        *(uint4*)(vdst + hs / 2 + d0) = *(const uint4*)(vrow + hs / 2 + d0);
    }
}

// The reference's stand-alone rotary kernel (pos_encoding_kernels.cu:39-77): in place on query and key, one thread per
// (token, head, rotation pair).
template <typename T, bool NEOX>
__global__ void rotary_embedding_kernel(const int64_t* __restrict__ positions, T* __restrict__ query, T* __restrict__ key,
                                        const T* __restrict__ cos_sin, int64_t cs_stride, int rot_dim, int64_t query_stride,
                                        int64_t key_stride, int num_heads, int num_kv_heads, int hs, int64_t num_tokens) {
#pragma clang fp contract(off)      // every product is rounded to the I/O dtype before the sum, as scalar_t arithmetic does in the reference
    using X = Tr<T>;
    const int e = rot_dim / 2;
    const int64_t per_tok = (int64_t)(num_heads + num_kv_heads) * e;
    const int64_t total = num_tokens * per_tok;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / per_tok;
        const int r = (int)(i % per_tok);
        const int head = r / e, ro = r % e;
        T* arr = head < num_heads ? query + t * query_stride + (int64_t)head * hs : key + t * key_stride + (int64_t)(head - num_heads) * hs;
        const T* cs = cos_sin + positions[t] * cs_stride;
        const int xi = NEOX ? ro : 2 * ro, yi = NEOX ? e + ro : 2 * ro + 1;
        const float cf = (float)cs[ro], sf = (float)cs[e + ro];
        const float xf = (float)arr[xi], yf = (float)arr[yi];
        const float p1 = (float)X::cvt(xf * cf), p2 = (float)X::cvt(yf * sf), q1 = (float)X::cvt(yf * cf), q2 = (float)X::cvt(xf * sf);
        arr[xi] = X::cvt(p1 - p2);
        arr[yi] = X::cvt(q1 + q2);
    }
}

// Append of k_new/v_new [b, sn, h_k, d] at row cache_seqlens[b] of slot cache_batch_idx[b]
// (flash_attn_interface.py:1168-1176).  16-byte chunks; d*itemsize is a multiple of 16.  With fused RoPE the key chunk of the
// first half and its partner chunk are rotated by the thread that owns the first-half chunk.
template <typename T> __device__ __forceinline__ void append_rope_pair(const vattn_attn_params& p, const uint16_t* src, uint16_t* dst, int c, int64_t pos) {
    using V8 = typename Tr<T>::v8;
    V8 x = as_v8<V8>(*(const uint4*)(src + c * 8)), y = as_v8<V8>(*(const uint4*)(src + p.d / 2 + c * 8));
    V8 cc, ss;
    rope_load<T>(p, pos, c * 8, cc, ss);
    rope8<T>(x, y, cc, ss);
    uint4 ux, uy;
    __builtin_memcpy(&ux, &x, 16);
    __builtin_memcpy(&uy, &y, 16);
    *(uint4*)(dst + c * 8) = ux;
    *(uint4*)(dst + p.d / 2 + c * 8) = uy;
}
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
        const uint16_t* ksrc = kn + b * p.knew_batch_stride + t * p.knew_row_stride + hk * p.knew_head_stride;
        uint16_t* kdst = kc + (int64_t)slot * p.k_batch_stride + (int64_t)row * p.k_row_stride + hk * p.k_head_stride;
        if (p.rotary_cos_sin) {
            if (c < cpr / 2) {
                if (p.dtype == VATTN_DTYPE_F16) append_rope_pair<_Float16>(p, ksrc, kdst, c, (int64_t)row);
                else append_rope_pair<__bf16>(p, ksrc, kdst, c, (int64_t)row);
            }
        } else {
            *(uint4*)(kdst + c * 8) = *(const uint4*)(ksrc + c * 8);
        }
        const uint4 vv = *(const uint4*)(vn + b * p.vnew_batch_stride + t * p.vnew_row_stride + hk * p.vnew_head_stride + c * 8);
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

int vattn_cache_flat_rope(const void* key, const void* value, void* k_cache, void* v_cache, int64_t num_tokens, int32_t num_heads,
                          int32_t head_size, int64_t key_stride, int64_t value_stride, int64_t k_cache_stride, int64_t v_cache_stride,
                          int32_t dtype, const void* cos_sin, int64_t cos_sin_row_stride, int64_t pos0, void* stream) {
    if (num_tokens <= 0) return VATTN_K_OK;
    if (!key || !value || !k_cache || !v_cache || !cos_sin) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    if (dtype != VATTN_DTYPE_F16 && dtype != VATTN_DTYPE_BF16) return fail(VATTN_K_ERR_UNSUPPORTED, "cache_flat_rope supports fp16 and bf16");
    if (head_size % 16 != 0 || ((key_stride | value_stride | k_cache_stride | v_cache_stride | cos_sin_row_stride) & 7) ||
        ((((uintptr_t)key) | ((uintptr_t)value) | ((uintptr_t)k_cache) | ((uintptr_t)v_cache) | ((uintptr_t)cos_sin)) & 15))
        return fail(VATTN_K_ERR_UNSUPPORTED, "cache_flat_rope needs 16-byte aligned rows (head_size a multiple of 16, strides of 8 elements)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = num_tokens * num_heads * (head_size / 16);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == VATTN_DTYPE_F16)
        hipLaunchKernelGGL(cache_flat_rope_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)key, (const _Float16*)value,
                           (_Float16*)k_cache, (_Float16*)v_cache, num_tokens, num_heads, head_size, key_stride, value_stride, k_cache_stride,
                           v_cache_stride, (const _Float16*)cos_sin, cos_sin_row_stride, pos0);
    else
        hipLaunchKernelGGL(cache_flat_rope_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)key, (const __bf16*)value,
                           (__bf16*)k_cache, (__bf16*)v_cache, num_tokens, num_heads, head_size, key_stride, value_stride, k_cache_stride,
                           v_cache_stride, (const __bf16*)cos_sin, cos_sin_row_stride, pos0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

int vattn_rotary_embedding(const int64_t* positions, void* query, void* key, int64_t num_tokens, int32_t num_q_heads, int32_t num_kv_heads,
                           int32_t head_size, int64_t query_stride, int64_t key_stride, int32_t dtype, const void* cos_sin,
                           int64_t cos_sin_row_stride, int32_t rot_dim, int32_t is_neox, void* stream) {
    if (num_tokens <= 0) return VATTN_K_OK;
    if (!positions || !query || !key || !cos_sin) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    if (dtype != VATTN_DTYPE_F16 && dtype != VATTN_DTYPE_BF16) return fail(VATTN_K_ERR_UNSUPPORTED, "rotary_embedding supports fp16 and bf16");
    if (rot_dim <= 0 || rot_dim > head_size || (rot_dim & 1)) return fail(VATTN_K_ERR_INVALID, "bad rotary dimension");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = num_tokens * (int64_t)(num_q_heads + num_kv_heads) * (rot_dim / 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
#define VATTN_ROT(TT, NX)                                                                                                                   \
    hipLaunchKernelGGL((rotary_embedding_kernel<TT, NX>), dim3((unsigned)blocks), dim3(256), 0, st, positions, (TT*)query, (TT*)key,        \
                       (const TT*)cos_sin, cos_sin_row_stride, rot_dim, query_stride, key_stride, num_q_heads, num_kv_heads, head_size, num_tokens)
    if (dtype == VATTN_DTYPE_F16) { if (is_neox) VATTN_ROT(_Float16, true); else VATTN_ROT(_Float16, false); }
    else { if (is_neox) VATTN_ROT(__bf16, true); else VATTN_ROT(__bf16, false); }
#undef VATTN_ROT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

}  // extern "C"
