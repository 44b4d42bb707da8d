// Shared types and device helpers of the gfx950 attention kernels (see attn_api.hip for the overview).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/vattn_kernels.h"

namespace vattn_k {


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr float kLog2e = 1.4426950408889634f;

template <typename T> struct Tr;
template <> struct Tr<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ _Float16 cvt(float x) { return (_Float16)x; }
};
template <> struct Tr<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ __bf16 cvt(float x) { return (__bf16)x; }
};

template <typename V8> __device__ __forceinline__ V8 as_v8(uint4 x) {
    V8 r;
    __builtin_memcpy(&r, &x, 16);
    return r;
}
template <typename V8> __device__ __forceinline__ V8 join_tr(s16x4 lo, s16x4 hi) {
    s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    V8 r;
    __builtin_memcpy(&r, &t, 16);
    return r;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float xor_shuffle(float v, int mask) { return __shfl_xor(v, mask, 64); }
// value of lane (l ^ 32): one v_permlane32_swap (VALU) instead of a ds_bpermute round trip through LDS
__device__ __forceinline__ float swap_halves(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = {lo,lo}, r[1] = {hi,hi}
    const unsigned other = (threadIdx.x & 32) ? r[0] : r[1];
    return __builtin_bit_cast(float, other);
}

// Bounds-checked 16-byte loads through a buffer descriptor: a lane whose byte offset lies at or beyond
// `bytes` gets zeros WITHOUT touching memory, so rows past the sequence's visible length (possibly on
// unmapped virtual pages) are never accessed, and the load stream is branch-free (counted vmcnt waits).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    // the byte count comes out of a clamp that instruction selection turns into a VALU v_med3: pull it
    // back into an SGPR, otherwise every load is wrapped in a waterfall loop
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
// wave-uniform pointer: make the uniformity provable so the descriptor lives in SGPRs (no waterfall loop)
template <typename P> __device__ __forceinline__ const P* uniform_ptr(const P* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (const P*)(((unsigned long long)hi << 32) | lo);
}

// ---- rotary position embedding, NeoX pairing (element i with element i + rot_dim/2), on 8-element fragments ----
// Arithmetic of /root/reference/sarathi-lean/csrc/pos_encoding_kernels.cu:32-35 in `scalar_t`: x' = x*cos - y*sin, y' = y*cos + x*sin
// with every product and the sum rounded to the I/O dtype (restated by oracle/attn.py rotary_embedding_ref; bit-exact).
template <typename T> __device__ __forceinline__ void rope8(typename Tr<T>::v8& x, typename Tr<T>::v8& y, typename Tr<T>::v8 c, typename Tr<T>::v8 s) {
#pragma clang fp contract(off)      // x*c - y*s must NOT become an fma: the reference rounds both products first (bit-exact K rows)
    using X = Tr<T>;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float xf = (float)x[j], yf = (float)y[j], cf = (float)c[j], sf = (float)s[j];
        const float p1 = (float)X::cvt(xf * cf), p2 = (float)X::cvt(yf * sf);
        const float q1 = (float)X::cvt(yf * cf), q2 = (float)X::cvt(xf * sf);
        x[j] = X::cvt(p1 - p2);
        y[j] = X::cvt(q1 + q2);
    }
}
// cos / sin fragments of row `pos` for elements [d0, d0 + 8) of the first half
template <typename T> __device__ __forceinline__ void rope_load(const vattn_attn_params& p, int64_t pos, int d0, typename Tr<T>::v8& c, typename Tr<T>::v8& s) {
    const T* row = (const T*)p.rotary_cos_sin + pos * p.rotary_row_stride;
    c = as_v8<typename Tr<T>::v8>(*(const uint4*)(row + d0));
    s = as_v8<typename Tr<T>::v8>(*(const uint4*)(row + p.rotary_dim / 2 + d0));
}

// ---- prefill form: pieces shared by prefill_kernels.hip and prefill64_kernels.hip ----
constexpr int PF_BN = 64;              // keys per tile
template <int HD> struct PfSmem {
    static constexpr int kRowBytes = HD * 2;
    static constexpr int kTileBytes = PF_BN * HD * 2;           // K tile == V tile size
    static constexpr int kBufBytes = 2 * kTileBytes;            // K + V
    static constexpr int kTotal = 2 * kBufBytes;                // double buffered
    static constexpr int kVSubBytes = PF_BN * 64;               // one [64 keys][32 d] sub-tile
};

// Workgroup -> (batch entry, head, query block).  The dispatcher hands consecutive workgroup ids to consecutive XCDs
// (id & 7), each with its own L2, and starts them in id order.  order 2 (default) makes every XCD stream ONE kv head (its
// L2 then holds a single K/V stream that the G query heads x neighbouring query blocks running there share) and walks the
// query blocks heaviest-first across ALL heads, so the workgroups running at any time have near-equal lengths and move
// down K/V in step.  order 1: heaviest-first across heads without the XCD grouping.  order 0: grid (query block, head,
// batch) - block-major per head (its tail is one head's heaviest blocks: 20-40 % slower on whole-prompt shapes).
// Returns false for the padding workgroups of the 1-D grids.
// KV-split (nsplit > 1, 1-D grids only): the grid is nsplit times larger; every run of 8 consecutive base ids (one per XCD) is
// repeated nsplit times, so the splits of a work item stay on its XCD and start together.
__device__ __forceinline__ bool wg_to_work(const vattn_attn_params& p, int order, int nqb, int nsplit, int& b, int& h, int& qb, int& split) {
    split = 0;
    if (order == 0) {
        b = blockIdx.z; h = blockIdx.y; qb = (int)gridDim.x - 1 - (int)blockIdx.x;
        return true;
    }
    int L = blockIdx.x;
    if (nsplit > 1) {
        const int grp = L >> 3;
        split = grp % nsplit;
        L = ((grp / nsplit) << 3) | (L & 7);
    }
    const int G = p.h / p.h_k;
    if (order == 2) {
        const int per = 8 / p.h_k;                         // XCDs per kv head (launch guarantees 8 % h_k == 0)
        const int xcd = L & 7;
        int t = (L >> 3) * per + xcd / p.h_k;
        const int g = t % G; t /= G;
        b = t % p.b;
        const int qbr = t / p.b;
        if (qbr >= nqb) return false;
        h = (xcd % p.h_k) * G + g;
        qb = nqb - 1 - qbr;
        return true;
    }
    h = L % p.h;
    const int t = L / p.h;
    b = t % p.b;
    qb = nqb - 1 - t / p.b;
    return t / p.b < nqb;
}


// ---- host-side pieces shared by the translation units ----
int fail(int code, const char* msg);                                    // attn_api.hip: records the message for vattn_kernels_last_error
void launch_append(const vattn_attn_params* p, hipStream_t st);         // cache_kernels.hip
int launch_prefill_form(const vattn_attn_params* p, hipStream_t st);    // prefill_kernels.hip (seqlen_q > 1)
size_t prefill_workspace_bytes(const vattn_attn_params* p);
void launch_prefill64(const vattn_attn_params* p, hipStream_t st, int nsplit);   // prefill64_kernels.hip (d = 128)
int launch_decode_form(const vattn_attn_params* p, hipStream_t st);     // decode_kernels.hip (seqlen_q == 1)
size_t decode_workspace_bytes(const vattn_attn_params* p);
int launch_hybrid(const vattn_attn_params* prefill, const vattn_attn_params* decode, void* ws, hipStream_t st);   // hybrid_kernels.hip
size_t hybrid_workspace_bytes(const vattn_attn_params* prefill, const vattn_attn_params* decode);

}  // namespace vattn_k
