// gfx950 (CDNA4) attention kernels over virtually-contiguous KV tensors.
//
//   prefill_kernel : chunked causal prefill (seqlen_q > 1).  One workgroup = 4 waves x 32 query rows,
//                    KV tiles of 64 keys double-buffered in LDS (register-staged global loads issued
//                    one tile ahead), S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_{f16,bf16},
//                    softmax entirely in registers (the "swapped QK^T" form: a lane owns one query
//                    column), K tile XOR-swizzled for conflict-free ds_read_b128, V tile stored as
//                    [d-block][key][32 d] sub-tiles and consumed through ds_read_b64_tr_b16.
//   decode_kernel  : seqlen_q == 1, GQA group packed into the MFMA N dimension, split-KV over the
//                    context, K fragments loaded straight from HBM into MFMA operand registers,
//                    V through a wave-private LDS transpose stage, fp32 online softmax, in-workgroup
//                    merge of the 4 waves, LSE-weighted combine across splits (combine_kernel).
//   cache_flat / append : contiguous KV append (16-byte vector copies).
//
// Semantics follow the operator the reference calls (flash_attn_with_kvcache):
//   /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, flash_api.cpp:1291-1578,
//   mask.h:164-196 (bottom-right causal), softmax.h:69-157 (fp32 max/sum, exp2, P rounded to the
//   I/O dtype before PV), flash_fwd_kernel.h:1116-1297 (split combine).
// Every K/V access is predicated on the sequence's visible length: rows at or beyond it may sit
// on unmapped virtual pages (SURVEY §7 "never touch unmapped VA").
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/vattn_kernels.h"

namespace {

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

// ============================================================================================
// prefill
// ============================================================================================

constexpr int PF_BN = 64;              // keys per tile
// Debug-only ablation switch for tools/kbench.py (cdna guide §5.4: "ablate before optimizing"); the product
// build leaves it at 0.  1: exp2 replaced by a multiply; 2: V^T fragments not read from LDS; 3: K fragments
// not read from LDS; 4: no global loads / LDS stores of the next tile; 5: no per-tile barrier; 6: no softmax VALU at all
#ifndef VATTN_ABLATE
#define VATTN_ABLATE 0
#endif
#ifndef VATTN_ABLATE_MASK
#define VATTN_ABLATE_MASK (VATTN_ABLATE ? (1 << VATTN_ABLATE) : 0)
#endif
#define ABL(k) ((VATTN_ABLATE_MASK >> (k)) & 1)

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

// WAVES waves per workgroup, each owning QC blocks of 32 query rows (BM = 32*QC*WAVES rows per workgroup).
// QC = 2 halves the LDS fragment traffic per flop (each K / V^T fragment read feeds two MFMAs) at the price
// of a 512-register budget (one wave per SIMD).
// MSUM: the softmax denominator is accumulated by the matrix pipe (one extra MFMA per 16 keys with an all-ones A
// fragment, no LDS read) instead of 32 dependent v_add per tile: the kernel is VALU/issue-bound, the matrix pipe has slack.
template <typename T, int HD, bool USE_TR, int WAVES, int QC, bool MSUM>
__global__ __launch_bounds__(64 * WAVES, (QC == 2 || HD > 128) ? 1 : 2) void prefill_kernel(vattn_attn_params p, int order, int nqb, int nsplit) {
    using X = Tr<T>;
    using V8 = typename X::v8;
    using S = PfSmem<HD>;
    constexpr int NT = 64 * WAVES;
    constexpr int BM = 32 * QC * WAVES;
    constexpr int KK = HD / 16;        // k-steps of the S^T MFMA chain
    constexpr int DB = HD / 32;        // 32-wide d blocks of O^T
    constexpr int CPR = HD / 8;        // 16-byte chunks per K/V row
    constexpr int PASSES = (PF_BN * CPR) / NT;
    constexpr int SWZ = CPR < 16 ? CPR - 1 : 15;   // K-tile swizzle mask
    static_assert(PASSES >= 1 && (PF_BN * CPR) % NT == 0, "tile does not divide over the workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    int b, h, qb, split;
    if (!wg_to_work(p, order, nqb, nsplit, b, h, qb, split)) return;
    const int hk = h / (p.h / p.h_k);                          // GQA: head h uses kv head h / (Hq/Hkv)
    // loaded values are wave-uniform; readfirstlane makes that provable (descriptors must live in SGPRs)
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    const int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
    // batched chunks of different lengths: entry b owns rows [q_first, q_first + Sq) of the flattened q / out
    const int Sq = p.q_lens ? __builtin_amdgcn_readfirstlane(p.q_lens[b]) : p.seqlen_q;
    const int64_t q_first = p.q_start ? (int64_t)__builtin_amdgcn_readfirstlane(p.q_start[b]) : 0;
    const bool causal = p.is_causal != 0;
    const int off = Lk - Sq;                                   // bottom-right alignment (mask.h:164-196)
    const int q_wg0 = qb * BM;
    if (q_wg0 >= Sq) return;                                   // shorter chunk than the grid was sized for (before any barrier)
    const int qw0 = q_wg0 + wave * 32 * QC;                    // first query row of this wave

    int n_end = Lk;
    if (causal) n_end = min(Lk, q_wg0 + BM + off);             // last key any row of this block may see, +1
    if (n_end < 0) n_end = 0;
    const int nt_all = (n_end + PF_BN - 1) / PF_BN;
    // KV-split: this workgroup owns key tiles [tb, nt) of the block's nt_all (an even share; shares past the end are empty
    // and fall through to the epilogue, which then publishes a zero partial with lse = -inf)
    int tb = 0, nt = nt_all;
    if (nsplit > 1) {
        const int per = (nt_all + nsplit - 1) / nsplit;
        tb = min(nt_all, split * per);
        nt = min(nt_all, tb + per);
    }

    const T* kbase = (const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride;
    const T* vbase = (const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride;

    // ---- Q^T fragments (B operand of S^T = K.Q^T): slot (g, j) <-> d = 16*kk + 8*g + j ----
    V8 qf[QC][KK];
#pragma unroll
    for (int qc = 0; qc < QC; qc++) {
        const int my_q = qw0 + 32 * qc + l31;
        const T* qptr = (const T*)p.q + (p.q_start ? 0 : (int64_t)b * p.q_batch_stride) + (q_first + my_q) * p.q_row_stride + (int64_t)h * p.q_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (my_q < Sq) v = *(const uint4*)(qptr + 16 * kk + 8 * g);
            qf[qc][kk] = as_v8<V8>(v);
        }
    }
    // Retire the Q loads HERE and make that visible to hipcc's wait-count pass: otherwise it keeps a conservative
    // "Q may still be in flight" state around the loop and puts a vmcnt wait in front of the first MFMA of every
    // tile, which also drains the K/V prefetch issued a moment earlier (vmcnt(0) = 0x0F70: expcnt/lgkmcnt untouched).
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int qc = 0; qc < QC; qc++)
#pragma unroll
        for (int kk = 0; kk < KK; kk++) asm volatile("" : "+v"(qf[qc][kk]));

    f32x16 o[DB][QC];
#pragma unroll
    for (int i = 0; i < DB; i++)
#pragma unroll
        for (int qc = 0; qc < QC; qc++) o[i][qc] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float m_run[QC], l_run[QC];   // running max of raw scores (same in both half-lanes); lane-local partial sums
    f32x16 lacc[QC];              // MSUM: every row of this accumulator holds the query's running denominator
    V8 ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = X::cvt(1.0f);
#pragma unroll
    for (int qc = 0; qc < QC; qc++) {
        m_run[qc] = -INFINITY;
        l_run[qc] = 0.f;
        lacc[qc] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    const float sc = p.softmax_scale * kLog2e;

    // two register sets: the loads of tile t+2 are issued while tile t is computed and are only consumed (stored to
    // LDS) at the end of iteration t+1 -> a two-iteration latency budget instead of one (L2/MALL latency under load
    // is about one tile time)
    uint4 kregA[PASSES], vregA[PASSES], kregB[PASSES], vregB[PASSES];
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;
    // per-thread byte offsets inside a tile (row-major rows of the cache, 16-byte chunk c)
    unsigned koff[PASSES], voff[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int idx = ps * NT + tid;
        koff[ps] = (unsigned)(idx / CPR) * k_rs_bytes + (unsigned)(idx % CPR) * 16u;
        voff[ps] = (unsigned)(idx / CPR) * v_rs_bytes + (unsigned)(idx % CPR) * 16u;
    }
    const T* kbase_u = uniform_ptr(kbase);
    const T* vbase_u = uniform_ptr(vbase);
    auto stage_load = [&](int t, uint4 (&kreg)[PASSES], uint4 (&vreg)[PASSES]) {
        // descriptor rebased per tile: rows at or beyond Lk fall outside num_records -> zeros, no access
        int rem = Lk - t * PF_BN;
        rem = rem < 0 ? 0 : (rem > PF_BN ? PF_BN : rem);
        const __amdgpu_buffer_rsrc_t kr = make_rsrc(kbase_u + (int64_t)t * PF_BN * p.k_row_stride, (unsigned)rem * k_rs_bytes);
        const __amdgpu_buffer_rsrc_t vr = make_rsrc(vbase_u + (int64_t)t * PF_BN * p.v_row_stride, (unsigned)rem * v_rs_bytes);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            kreg[ps] = buf_load16(kr, koff[ps]);
            vreg[ps] = buf_load16(vr, voff[ps]);
        }
    };
    auto stage_write = [&](int buf, const uint4 (&kreg)[PASSES], const uint4 (&vreg)[PASSES]) {
        char* ksm = smem + buf * S::kBufBytes;
        char* vsm = ksm + S::kTileBytes;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int idx = ps * NT + tid;
            const int row = idx / CPR;
            const int c = idx % CPR;
            // K: row-major, 16-byte chunk index XOR-swizzled with (row & 15) -> conflict-free ds_read_b128
            // (d = 64: 8 chunks per row, swizzle with row & 7)
            *(uint4*)(ksm + row * S::kRowBytes + ((c ^ (row & SWZ)) << 4)) = kreg[ps];
            // V: [d/32][key][32 d] sub-tiles (64-byte rows) for the transpose reads
            *(uint4*)(vsm + (c >> 2) * S::kVSubBytes + row * 64 + ((c & 3) << 4)) = vreg[ps];
        }
    };

    if (nt > tb) {
        stage_load(tb, kregA, vregA);
        stage_write(0, kregA, vregA);
        stage_load(tb + 1, kregB, vregB);
    }
    __syncthreads();

    auto tile_body = [&](int t, uint4 (&kld)[PASSES], uint4 (&vld)[PASSES], const uint4 (&kwr)[PASSES], const uint4 (&vwr)[PASSES]) {
        const int buf = (t - tb) & 1;
        if (!ABL(4)) stage_load(t + 2, kld, vld);     // two tiles ahead (past the last tile: all lanes out of range)

        const int n0 = t * PF_BN;
        // wave-uniform tile classification
        const bool wave_dead = causal && (n0 > qw0 + 32 * QC - 1 + off);          // every (row, key) pair masked
        if (!wave_dead) {
            const char* ksm = smem + buf * S::kBufBytes;
            const char* vsm = ksm + S::kTileBytes;
            f32x16 s[2][QC];
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int qc = 0; qc < QC; qc++) s[kb][qc] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // k-step outer, key-block inner: consecutive MFMAs hit DIFFERENT accumulators, so the dependent
            // accumulate latency of one chain is covered by the other chain's issue slot; the K fragments of
            // step kk+1 are read from LDS while the MFMAs of step kk run (explicit two-deep register ring)
            auto kfrag = [&](int kb, int kk) -> V8 {
                if (ABL(3)) return qf[0][(kk + kb) % KK];
                return *(const V8*)(ksm + (kb * 32 + l31) * S::kRowBytes + (((2 * kk + g) ^ (l31 & SWZ)) << 4));
            };
            V8 a_cur[2], a_nxt[2];
            a_cur[0] = kfrag(0, 0);
            a_cur[1] = kfrag(1, 0);
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                if (kk + 1 < KK) {
                    a_nxt[0] = kfrag(0, kk + 1);
                    a_nxt[1] = kfrag(1, kk + 1);
                }
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int qc = 0; qc < QC; qc++) s[kb][qc] = X::mfma32(a_cur[kb], qf[qc][kk], s[kb][qc]);
                a_cur[0] = a_nxt[0];
                a_cur[1] = a_nxt[1];
            }
            // pin the issue order the ring is meant to have (hipcc otherwise sinks every read next to its use):
            // reads of step kk+1, then the MFMAs of step kk   (LLVM SchedGroupMask: 0x100 = DS read, 0x8 = MFMA)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int kk = 0; kk + 1 < KK; kk++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * QC, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * QC, 0);
            // s[kb][qc][r] = S^T[key = n0 + 32*kb + 8*(r>>2) + 4*g + (r&3)][query = qw0 + 32*qc + l31]
            const bool need_mask = (n0 + PF_BN > Lk) || (causal && (n0 + PF_BN - 1 > qw0 + off));
            float alpha[QC];
#pragma unroll
            for (int qc = 0; qc < QC; qc++) {
                if (ABL(6)) { alpha[qc] = 1.f; continue; }
                if (need_mask) {
                    const int my_q = qw0 + 32 * qc + l31;
                    const int lim = causal ? min(Lk - 1, my_q + off) : Lk - 1;     // last visible key for this query
#pragma unroll
                    for (int kb = 0; kb < 2; kb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int key = n0 + 32 * kb + 8 * (r >> 2) + 4 * g + (r & 3);
                            if (key > lim) s[kb][qc][r] = -INFINITY;
                        }
                }
                float mloc = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) mloc = fmaxf(mloc, s[kb][qc][r]);
                mloc = fmaxf(mloc, swap_halves(mloc));
                const float m_new = fmaxf(m_run[qc], mloc);
                const float msub = (m_new == -INFINITY) ? 0.f : m_new * sc;   // softmax.h: all-masked rows use 0
                alpha[qc] = fast_exp2(m_run[qc] * sc - msub);                  // m_run = -inf -> 0
                m_run[qc] = m_new;
                float psum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float e;
                        if (ABL(1)) e = s[kb][qc][r] * sc; else e = fast_exp2(__builtin_fmaf(s[kb][qc][r], sc, -msub));
                        s[kb][qc][r] = e;
                        if (!MSUM) psum += e;
                    }
                if (!MSUM) l_run[qc] = l_run[qc] * alpha[qc] + psum;
                // O only needs rescaling when some row's running max actually moved (rare after the first
                // tiles); the test is exact (alpha == 1 otherwise) and wave-uniform
                if (__builtin_amdgcn_ballot_w64(alpha[qc] != 1.0f) != 0) {
#pragma unroll
                    for (int i = 0; i < DB; i++)
#pragma unroll
                        for (int r = 0; r < 16; r++) o[i][qc][r] *= alpha[qc];
                    if (MSUM) {
#pragma unroll
                        for (int r = 0; r < 16; r++) lacc[qc][r] *= alpha[qc];
                    }
                }
            }

            // O^T += V^T . P^T : B operand slot (g, j) <-> key 16*u + (j<4 ? 4g+j : 8+4g+j-4) = S^T regs 8u..8u+7
            // LLVM's MFMA/exp interleaving strategy for this scheduling region: +0.7..2 % measured (937 -> 946 TF on the 32 k
            // prompt, 986 -> 999 on 4 k chunks); strategies 0 / 1 (small-GEMM interleaves) lose 0.5 %
            __builtin_amdgcn_iglp_opt(2);
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    V8 pf[QC];
#pragma unroll
                    for (int qc = 0; qc < QC; qc++)
#pragma unroll
                        for (int j = 0; j < 8; j++) pf[qc][j] = X::cvt(s[kb][qc][8 * u + j]);
                    const int krow0 = kb * 32 + 16 * u;
                    if (MSUM) {
#pragma unroll
                        for (int qc = 0; qc < QC; qc++) lacc[qc] = X::mfma32(ones, pf[qc], lacc[qc]);
                    }
#pragma unroll
                    for (int db = 0; db < DB; db++) {
                        V8 a;
                        if (ABL(2)) {
                            a = qf[0][(db + u + 2 * kb) % KK];
                        } else if constexpr (USE_TR) {
                            const int i16 = lane & 15, dh = (lane >> 4) & 1;
                            const char* a1 = vsm + db * S::kVSubBytes + (krow0 + 4 * g + (i16 >> 2)) * 64 + (16 * dh + 4 * (i16 & 3)) * 2;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1 + 8 * 64));
                            a = join_tr<V8>(lo, hi);
                        } else {
                            const T* vs = (const T*)(vsm + db * S::kVSubBytes);
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const int key = krow0 + (j < 4 ? 4 * g + j : 8 + 4 * g + (j - 4));
                                a[j] = vs[key * 32 + l31];
                            }
                        }
#pragma unroll
                        for (int qc = 0; qc < QC; qc++) o[db][qc] = X::mfma32(a, pf[qc], o[db][qc]);
                    }
                }
        }
        if (!ABL(4)) stage_write(buf ^ 1, kwr, vwr);   // tile t+1 (issued one iteration ago) into the buffer last read in iteration t-1
        if (!ABL(5) && !ABL(4)) __syncthreads();
    };
    for (int t = tb; t < nt; t += 2) {
        tile_body(t, kregA, vregA, kregB, vregB);
        if (t + 1 < nt) tile_body(t + 1, kregB, vregB, kregA, vregA);
    }

    // ---- epilogue: O^T[d = 32*db + 8*(r>>2) + 4*g + (r&3)][query] ----
#pragma unroll
    for (int qc = 0; qc < QC; qc++) {
        const int my_q = qw0 + 32 * qc + l31;
        const float l_tot = MSUM ? lacc[qc][0] : (l_run[qc] + swap_halves(l_run[qc]));
        const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
        if (my_q < Sq && nsplit > 1) {
            // KV-split: normalised fp32 partial + its log2-domain LSE; combine_kernel merges the nsplit partials of a row
            // workspace: float o_part[nsplit][B][Sq][H][HD]; float lse_part[nsplit][B][Sq][H]
            const int64_t row = (((int64_t)split * p.b + b) * p.seqlen_q + my_q) * p.h + h;
            float* opart = (float*)p.workspace + row * HD;
            float* lpart = (float*)p.workspace + (int64_t)nsplit * p.b * p.seqlen_q * p.h * HD;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int tq = 0; tq < 4; tq++) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = o[db][qc][4 * tq + e] * inv;
                    *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                }
            if (g == 0) lpart[row] = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_run[qc] * sc + __log2f(l_tot));
        } else if (my_q < Sq) {
            T* optr = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + my_q) * p.o_row_stride + (int64_t)h * p.o_head_stride;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int tq = 0; tq < 4; tq++) {
                    typename X::v4 w;
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = X::cvt(o[db][qc][4 * tq + e] * inv);
                    *(typename X::v4*)(optr + 32 * db + 8 * tq + 4 * g) = w;
                }
            if (p.softmax_lse && g == 0) {
                // natural-log LSE of scale*QK^T; +inf for fully masked rows (flash convention)
                const float lse = (l_tot == 0.f) ? INFINITY : (m_run[qc] * p.softmax_scale + __logf(l_tot));
                p.softmax_lse[((int64_t)b * p.h + h) * p.seqlen_q + my_q] = lse;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Interleaved, software-pipelined prefill (8 waves x 32 query rows, d = 128): S(t+1) = K(t+1).Q^T is accumulated while
// the softmax of tile t is evaluated, then P(t).V(t); K runs one tile ahead of V in LDS (iteration t reads K[(t+1)&1] and
// V[t&1] and stores K(t+2) -> K[t&1], V(t+1) -> V[(t+1)&1] before its single barrier).  The ISSUE ORDER is written out by
// hand instead of left to the scheduler: the loop body is a sequence of 32 groups, each
//     { one MFMA ; the LDS fragment read(s) for the MFMA two groups ahead ; a 3-6 instruction slice of VALU work }
// closed by a scheduling barrier, so every MFMA is followed by independent VALU of the SAME wave.  Measured on gfx950
// (tools/mfma_overlap_probe.cpp, tools/attn_skeleton_probe.cpp): VALU issued by the wave that owns the running MFMA
// hides almost completely (16 MFMA + 64 VALU: +5 %), VALU issued by the OTHER wave of the SIMD costs the matrix pipe
// about half its issue time.  VALU placement per tile (138 instructions for 32 MFMAs):
//     QK(t+1) MFMAs 0-15 : exp2 / row-sum / f16 pack of P(t) values 0-19          (ten pairs, ~4.4 per MFMA)
//     PV(t)   MFMAs 0-7  : P(t) values 20-31                                       (six pairs, ~5.3 per MFMA)
//     PV(t)   MFMAs 8-15 : running max of S(t+1) (v_max3 chain), then m / alpha    (~3 per MFMA)
// so the row max never sits on the critical path between two matrix phases.
// --------------------------------------------------------------------------------------------
#ifndef ILV_AHEAD
#define ILV_AHEAD 2
#endif
template <typename T>
__global__ __launch_bounds__(512, 2) void prefill_ilv_kernel(vattn_attn_params p, int order, int nqb) {
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int HD = 128;
    using S = PfSmem<HD>;
    constexpr int WAVES = 8, NT = 64 * WAVES, BM = 32 * WAVES;
    constexpr int KK = HD / 16, DB = HD / 32, CPR = HD / 8;
    constexpr int PASSES = (PF_BN * CPR) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ksm0 = smem;                          // K[0], K[1]
    char* const vsm0 = smem + 2 * S::kTileBytes;      // V[0], V[1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;
    int b, h, qb, split_unused;
    if (!wg_to_work(p, order, nqb, 1, b, h, qb, split_unused)) return;
    const int hk = h / (p.h / p.h_k);
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    const int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
    const int Sq = p.seqlen_q;
    const bool causal = p.is_causal != 0;
    const int off = Lk - Sq;
    const int q_wg0 = qb * BM;
    const int qw0 = q_wg0 + wave * 32;
    const int my_q = qw0 + l31;

    int n_end = Lk;
    if (causal) n_end = min(Lk, q_wg0 + BM + off);
    if (n_end < 0) n_end = 0;
    const int nt = (n_end + PF_BN - 1) / PF_BN;
    int t_live = nt;      // tiles [0, t_live) hold at least one visible (row, key) pair for THIS wave (wave-uniform)
    if (causal) {
        const int last_key = qw0 + 31 + off;
        t_live = last_key < 0 ? 0 : min(nt, last_key / PF_BN + 1);
    }

    const T* kbase = (const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride;
    const T* vbase = (const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride;
    const T* qptr = (const T*)p.q + (int64_t)b * p.q_batch_stride + (int64_t)my_q * p.q_row_stride + (int64_t)h * p.q_head_stride;

    V8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (my_q < Sq) v = *(const uint4*)(qptr + 16 * kk + 8 * g);
        qf[kk] = as_v8<V8>(v);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // retire the Q loads here (see prefill_kernel)
#pragma unroll
    for (int kk = 0; kk < KK; kk++) asm volatile("" : "+v"(qf[kk]));

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; i++) o[i] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.softmax_scale * kLog2e;

    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;
    unsigned koff[PASSES], voff[PASSES], klds[PASSES], vlds[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int idx = ps * NT + tid;
        const int row = idx / CPR, c = idx % CPR;
        koff[ps] = (unsigned)row * k_rs_bytes + (unsigned)c * 16u;
        voff[ps] = (unsigned)row * v_rs_bytes + (unsigned)c * 16u;
        klds[ps] = (unsigned)(row * S::kRowBytes + ((c ^ (row & 15)) << 4));                 // XOR-swizzled K image
        vlds[ps] = (unsigned)((c >> 2) * S::kVSubBytes + row * 64 + ((c & 3) << 4));         // [d/32][key][32 d] V image
    }
    const T* kbase_u = uniform_ptr(kbase);
    const T* vbase_u = uniform_ptr(vbase);
    auto tile_rsrc = [&](const T* base, int64_t row_stride, unsigned rs_bytes, int t) {
        int rem = Lk - t * PF_BN;
        rem = rem < 0 ? 0 : (rem > PF_BN ? PF_BN : rem);
        return make_rsrc(base + (int64_t)t * PF_BN * row_stride, (unsigned)rem * rs_bytes);
    };
    uint4 kreg[PASSES], vreg[PASSES];
    auto load_k = [&](int t) {
        const __amdgpu_buffer_rsrc_t r = tile_rsrc(kbase_u, p.k_row_stride, k_rs_bytes, t);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) kreg[ps] = buf_load16(r, koff[ps]);
    };
    auto load_v = [&](int t) {
        const __amdgpu_buffer_rsrc_t r = tile_rsrc(vbase_u, p.v_row_stride, v_rs_bytes, t);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) vreg[ps] = buf_load16(r, voff[ps]);
    };
    auto store_k = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) *(uint4*)(ksm0 + buf * S::kTileBytes + klds[ps]) = kreg[ps];
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) *(uint4*)(vsm0 + buf * S::kTileBytes + vlds[ps]) = vreg[ps];
    };
    // K fragment of S^T MFMA i (k-step i>>1, key block i&1); V^T fragment of PV MFMA j (P group j>>2, d block j&3)
    const unsigned kfrag_row = (unsigned)(l31 * S::kRowBytes);
    auto kfrag = [&](const char* ksm, int i) -> V8 {
        const int kk = i >> 1, kb = i & 1;
        return *(const V8*)(ksm + kb * 32 * S::kRowBytes + kfrag_row + (((2 * kk + g) ^ (l31 & 15)) << 4));
    };
    const int i16 = lane & 15, dh = (lane >> 4) & 1;
    const unsigned vfrag_lane = (unsigned)((4 * g + (i16 >> 2)) * 64 + (16 * dh + 4 * (i16 & 3)) * 2);
    auto vfrag = [&](const char* vsm, int j) -> V8 {
        const int pg = j >> 2, db = j & 3;
        const int krow0 = (pg >> 1) * 32 + 16 * (pg & 1);
        const char* a1 = vsm + db * S::kVSubBytes + krow0 * 64 + vfrag_lane;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1 + 8 * 64));
        return join_tr<V8>(lo, hi);
    };
    auto mask_tile = [&](int tt, f32x16& x0, f32x16& x1) {
        const int n0 = tt * PF_BN;
        if ((n0 + PF_BN > Lk) || (causal && (n0 + PF_BN - 1 > qw0 + off))) {
            const int lim = causal ? min(Lk - 1, my_q + off) : Lk - 1;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = n0 + 8 * (r >> 2) + 4 * g + (r & 3);
                if (key > lim) x0[r] = -INFINITY;
                if (key + 32 > lim) x1[r] = -INFINITY;
            }
        }
    };

    // ---- prologue: K(0), K(1), V(0) into LDS; S(0); its row max ----
    load_k(0);
    store_k(0);
    load_k(1);
    store_k(1);
    load_v(0);
    store_v(0);
    __syncthreads();
    f32x16 sc0 = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 sc1 = sc0;      // S(t): key blocks 0 and 1 of the current tile
    float msub = 0.f, alpha = 1.f;   // softmax shift of the current tile, rescale factor it implies for O and l
    if (t_live > 0) {
#pragma unroll
        for (int i = 0; i < 2 * KK; i++) {
            if (i & 1) sc1 = X::mfma32(kfrag(ksm0, i), qf[i >> 1], sc1);
            else sc0 = X::mfma32(kfrag(ksm0, i), qf[i >> 1], sc0);
        }
        mask_tile(0, sc0, sc1);
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) mloc = fmaxf(mloc, fmaxf(sc0[r], sc1[r]));
        mloc = fmaxf(mloc, swap_halves(mloc));
        m_run = mloc;
        msub = (m_run == -INFINITY) ? 0.f : m_run * sc;
    }

    for (int t = 0; t < nt; t++) {
        load_k(t + 2);      // in flight across the whole iteration (out of range past the end: zeros, no access)
        load_v(t + 1);
        if (t < t_live) {
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {      // rare after the first tiles; exact
#pragma unroll
                for (int i = 0; i < DB; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o[i][r] *= alpha;
            }
            const char* ksm = ksm0 + ((t + 1) & 1) * S::kTileBytes;
            const char* vsm = vsm0 + (t & 1) * S::kTileBytes;
            f32x16 sn0 = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 sn1 = sn0;
            V8 pf[4];            // P(t) in the PV B-operand layout: group pg <-> keys of S^T registers 8u..8u+7 of key block kb, pg = 2kb+u
            float psum = 0.f;
            // P pair pr (0..15): key block pr>>3, S^T registers 2*(pr&7), +1  ->  pf[pr>>2] elements 2*(pr&3), +1
            auto exp_pair = [&](int pr) {
                const int kb = pr >> 3, r0 = 2 * (pr & 7);
                const float e0 = fast_exp2(__builtin_fmaf(kb ? sc1[r0] : sc0[r0], sc, -msub));
                const float e1 = fast_exp2(__builtin_fmaf(kb ? sc1[r0 + 1] : sc0[r0 + 1], sc, -msub));
                psum += e0;
                psum += e1;
                pf[pr >> 2][2 * (pr & 3)] = X::cvt(e0);
                pf[pr >> 2][2 * (pr & 3) + 1] = X::cvt(e1);
            };
            constexpr int AH = ILV_AHEAD;      // fragment reads run AH groups ahead of the MFMA that consumes them
            V8 kf[2 * KK];
#pragma unroll
            for (int i = 0; i < AH; i++) kf[i] = kfrag(ksm, i);
            __builtin_amdgcn_sched_barrier(0);
            V8 vf[16];
            // ---- S(t+1) = K(t+1).Q^T   ||   P(t) pairs 0-9 ----
#pragma unroll
            for (int i = 0; i < 2 * KK; i++) {
                if (i + AH < 2 * KK) kf[i + AH] = kfrag(ksm, i + AH);
                else vf[i + AH - 2 * KK] = vfrag(vsm, i + AH - 2 * KK);     // the last AH groups prefetch the first V^T fragments
                if (i & 1) sn1 = X::mfma32(kf[i], qf[i >> 1], sn1);
                else sn0 = X::mfma32(kf[i], qf[i >> 1], sn0);
#pragma unroll
                for (int pr = (i * 10 + 15) / 16; pr < ((i + 1) * 10 + 15) / 16; pr++) exp_pair(pr);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- O^T += V^T(t).P(t)^T, first half   ||   P(t) pairs 10-15 ----
#pragma unroll
            for (int j = 0; j < 8; j++) {
                vf[j + AH] = vfrag(vsm, j + AH);
                o[j & 3] = X::mfma32(vf[j], pf[j >> 2], o[j & 3]);
#pragma unroll
                for (int pr = 10 + (j * 6 + 7) / 8; pr < 10 + ((j + 1) * 6 + 7) / 8; pr++) exp_pair(pr);
                __builtin_amdgcn_sched_barrier(0);
            }
            l_run = l_run * alpha + psum;
            // ---- second half   ||   row max of S(t+1) (raw: a diagonal / ragged next tile is redone below) ----
            // NOTE no control flow between the groups of one iteration: hipcc sinks the VALU slices of a block into a
            // later block when their results are only used there, across scheduling barriers
            float mx = -INFINITY;
#pragma unroll
            for (int j = 8; j < 16; j++) {
                if (j + AH < 16) vf[j + AH] = vfrag(vsm, j + AH);
                o[j & 3] = X::mfma32(vf[j], pf[j >> 2], o[j & 3]);
                {
                    const int q = j - 8;      // S(t+1) registers 2q, 2q+1 of both key blocks
                    mx = fmaxf(fmaxf(mx, sn0[2 * q]), sn0[2 * q + 1]);
                    mx = fmaxf(fmaxf(mx, sn1[2 * q]), sn1[2 * q + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool has_next = t + 1 < t_live;
            const int n1 = (t + 1) * PF_BN;
            if (has_next && ((n1 + PF_BN > Lk) || (causal && (n1 + PF_BN - 1 > qw0 + off)))) {     // wave-uniform, rare
                mask_tile(t + 1, sn0, sn1);
                mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r++) mx = fmaxf(mx, fmaxf(sn0[r], sn1[r]));
            }
            mx = fmaxf(mx, swap_halves(mx));
            const float m_new = has_next ? fmaxf(m_run, mx) : m_run;
            msub = (m_new == -INFINITY) ? 0.f : m_new * sc;
            alpha = fast_exp2(m_run * sc - msub);
            m_run = m_new;
            sc0 = sn0;
            sc1 = sn1;
        }
        store_k(t & 1);            // K(t+2): the buffer held K(t), whose QK finished in iteration t-1 on every wave
        store_v((t + 1) & 1);      // V(t+1): the buffer held V(t-1), last read in iteration t-1
        __syncthreads();
    }

    const float l_tot = l_run + swap_halves(l_run);
    const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
    if (my_q < Sq) {
        T* optr = (T*)p.out + (int64_t)b * p.o_batch_stride + (int64_t)my_q * p.o_row_stride + (int64_t)h * p.o_head_stride;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int tq = 0; tq < 4; tq++) {
                typename X::v4 w;
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = X::cvt(o[db][4 * tq + e] * inv);
                *(typename X::v4*)(optr + 32 * db + 8 * tq + 4 * g) = w;
            }
        if (p.softmax_lse && g == 0) {
            const float lse = (l_tot == 0.f) ? INFINITY : (m_run * p.softmax_scale + __logf(l_tot));
            p.softmax_lse[((int64_t)b * p.h + h) * Sq + my_q] = lse;
        }
    }
}

// ============================================================================================
// decode (seqlen_q == 1): split-KV
// ============================================================================================

constexpr int DC_WAVES = 4;
constexpr int DC_BN = 32;     // keys per wave tile

// workspace layout: float o_accum[splits][b][h][d]; float lse_accum[splits][b][h]  (log2 domain, scaled)
template <typename T, int HD, bool USE_TR>
__global__ __launch_bounds__(64 * DC_WAVES, HD > 128 ? 2 : 3) void decode_kernel(vattn_attn_params p, int num_splits, int gblocks, int fused_append) {
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int KK = HD / 32;          // k-steps of S^T (16x16x32)
    constexpr int DB = HD / 16;          // 16-wide d blocks of O^T
    constexpr int CPR = HD / 8;          // 16-byte chunks per row
    constexpr int VPASS = (DC_BN * CPR) / 64;
    constexpr int V_WAVE_BYTES = DC_BN * HD * 2;        // [d/16][32 keys][16 d] sub-tiles, 32-byte rows
    constexpr int VSUB = DC_BN * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int g4 = lane >> 4;

    const int split = blockIdx.x;
    const int hk = blockIdx.y / gblocks;
    const int gb = blockIdx.y % gblocks;
    const int b = blockIdx.z;
    const int G = p.h / p.h_k;
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    const int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);

    // each sequence divides ITS OWN length evenly over the splits (balanced for ragged batches)
    const int ntiles_total = (Lk + DC_BN - 1) / DC_BN;
    const int tiles_per_split = (ntiles_total + num_splits - 1) / num_splits;
    const int tile_begin = split * tiles_per_split;
    const int tile_end = min(ntiles_total, tile_begin + tiles_per_split);

    // Fused append (seqlen_knew == 1): the new K/V row sits at key index Lk-1.  Every workgroup that reads the tile
    // holding it substitutes the row from k_new/v_new in registers; the gb == 0 workgroup also stores it into the
    // cache (flash_attn_interface.py:1168-1176: append, then attend).  No inter-workgroup ordering is needed.
    const int new_key = fused_append ? Lk - 1 : -1;
    const int new_tile = fused_append ? new_key / DC_BN : -1;

    const int row_head = gb * 16 + l15;                 // query head within the group handled by this lane's column
    const bool row_valid = row_head < G;
    const int h = hk * G + row_head;
    const T* qptr = (const T*)p.q + (int64_t)b * p.q_batch_stride + (int64_t)h * p.q_head_stride;
    const T* kbase = (const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride;
    const T* vbase = (const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride;

    // Q^T fragments (B operand, n = query head): slot (g4, j) <-> d = 32*kk + 8*g4 + j
    V8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row_valid) v = *(const uint4*)(qptr + 32 * kk + 8 * g4);
        qf[kk] = as_v8<V8>(v);
    }

    f32x4 o[DB];
#pragma unroll
    for (int i = 0; i < DB; i++) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.softmax_scale * kLog2e;
    char* vsm = smem + wave * V_WAVE_BYTES;

    uint4 kreg[2][KK], vreg[VPASS];
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;
    const T* kbase_u = uniform_ptr(kbase);
    const T* vbase_u = uniform_ptr(vbase);
    unsigned koff[2], voff[VPASS];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) koff[kb] = (unsigned)(16 * kb + l15) * k_rs_bytes + (unsigned)g4 * 16u;
#pragma unroll
    for (int ps = 0; ps < VPASS; ps++) {
        const int idx = ps * 64 + lane;
        voff[ps] = (unsigned)(idx / CPR) * v_rs_bytes + (unsigned)(idx % CPR) * 16u;
    }
    auto load_tile = [&](int tile) {
        const int k0 = tile * DC_BN;
        int rem = Lk - k0;
        rem = rem < 0 ? 0 : (rem > DC_BN ? DC_BN : rem);
        const __amdgpu_buffer_rsrc_t kr = make_rsrc(kbase_u + (int64_t)k0 * p.k_row_stride, (unsigned)rem * k_rs_bytes);
        const __amdgpu_buffer_rsrc_t vr = make_rsrc(vbase_u + (int64_t)k0 * p.v_row_stride, (unsigned)rem * v_rs_bytes);
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int kk = 0; kk < KK; kk++) kreg[kb][kk] = buf_load16(kr, koff[kb] + 64u * kk);
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) vreg[ps] = buf_load16(vr, voff[ps]);
        if (tile == new_tile) {      // wave-uniform, at most once per workgroup
            const T* kn = (const T*)p.k_new + (int64_t)b * p.knew_batch_stride + (int64_t)hk * p.knew_head_stride;
            const T* vn = (const T*)p.v_new + (int64_t)b * p.vnew_batch_stride + (int64_t)hk * p.vnew_head_stride;
            T* kc = (T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride + (int64_t)new_key * p.k_row_stride;
            T* vc = (T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride + (int64_t)new_key * p.v_row_stride;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
                if (k0 + 16 * kb + l15 == new_key) {
#pragma unroll
                    for (int kk = 0; kk < KK; kk++) {
                        const uint4 v = *(const uint4*)(kn + 32 * kk + 8 * g4);
                        kreg[kb][kk] = v;
                        if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(kc + 32 * kk + 8 * g4) = v;
                    }
                }
#pragma unroll
            for (int ps = 0; ps < VPASS; ps++) {
                const int idx = ps * 64 + lane;
                if (k0 + idx / CPR == new_key) {
                    const uint4 v = *(const uint4*)(vn + (idx % CPR) * 8);
                    vreg[ps] = v;
                    if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(vc + (idx % CPR) * 8) = v;
                }
            }
        }
    };

    int tile = __builtin_amdgcn_readfirstlane(tile_begin + wave);
    load_tile(tile < tile_end ? tile : ntiles_total);     // past the end: every lane out of range, no access
    for (; tile < tile_end; tile += DC_WAVES) {
        const int k0 = tile * DC_BN;
        // ---- V: registers -> wave-private LDS ([d/16][key][16 d]) ----
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) {
            const int idx = ps * 64 + lane;
            const int row = idx / CPR, c = idx % CPR;
            *(uint4*)(vsm + (c >> 1) * VSUB + row * 32 + ((c & 1) << 4)) = vreg[ps];
        }
        // ---- S^T = K.Q^T on the register-resident K fragments ----
        f32x4 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; kk++) s[kb] = X::mfma16(as_v8<V8>(kreg[kb][kk]), qf[kk], s[kb]);
        }
        // prefetch the wave's next tile while this one is being consumed (out of range past the split's end)
        load_tile(tile + DC_WAVES < tile_end ? tile + DC_WAVES : ntiles_total);

        // s[kb][r] = S^T[key = k0 + 16*kb + 4*g4 + r][head row l15]
        if (k0 + DC_BN > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (k0 + 16 * kb + 4 * g4 + r >= Lk) s[kb][r] = -INFINITY;
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) mloc = fmaxf(mloc, s[kb][r]);
        mloc = fmaxf(mloc, xor_shuffle(mloc, 16));
        mloc = fmaxf(mloc, xor_shuffle(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float msub = (m_new == -INFINITY) ? 0.f : m_new * sc;
        const float alpha = fast_exp2(m_run * sc - msub);
        m_run = m_new;
        float psum = 0.f;
        V8 pf;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = fast_exp2(__builtin_fmaf(s[kb][r], sc, -msub));
                psum += e;
                pf[4 * kb + r] = X::cvt(e);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[i][r] *= alpha;

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- O^T += V^T.P^T : A slot (g4, j) <-> key k0 + (j<4 ? 4*g4 + j : 16 + 4*g4 + j-4) ----
#pragma unroll
        for (int db = 0; db < DB; db++) {
            V8 a;
            if constexpr (USE_TR) {
                const char* a1 = vsm + db * VSUB + (4 * g4 + (l15 >> 2)) * 32 + (4 * (l15 & 3)) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1 + 16 * 32));
                a = join_tr<V8>(lo, hi);
            } else {
                const T* vs = (const T*)(vsm + db * VSUB);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int key = (j < 4) ? 4 * g4 + j : 16 + 4 * g4 + (j - 4);
                    a[j] = vs[key * 16 + l15];
                }
            }
            o[db] = X::mfma16(a, pf, o[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the 4 waves (each holds a partial softmax over its own tiles) ----
    l_run += xor_shuffle(l_run, 16);
    l_run += xor_shuffle(l_run, 32);
    __syncthreads();                                    // all waves are done with their V staging area
    // o[db][r] = O^T[d = 16*db + 4*g4 + r][head row l15]
    float* osm = (float*)smem;                          // [wave][16 rows][HD]
    float* msm = (float*)(smem + DC_WAVES * 16 * HD * 4);   // [wave][16] m, then [wave][16] l
    float* lsm = msm + DC_WAVES * 16;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
        for (int r = 0; r < 4; r++) osm[(wave * 16 + l15) * HD + 16 * db + 4 * g4 + r] = o[db][r];
    if (g4 == 0) {
        msm[wave * 16 + l15] = m_run;
        lsm[wave * 16 + l15] = l_run;
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * HD; idx += 64 * DC_WAVES) {
        const int row = idx / HD, d = idx % HD;
        const int rh = gb * 16 + row;
        if (rh >= G) continue;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < DC_WAVES; w++) mx = fmaxf(mx, msm[w * 16 + row]);
        float acc = 0.f, lsum = 0.f;
        const float mxs = (mx == -INFINITY) ? 0.f : mx * sc;
#pragma unroll
        for (int w = 0; w < DC_WAVES; w++) {
            const float f = fast_exp2(msm[w * 16 + row] * sc - mxs);
            acc += f * osm[(w * 16 + row) * HD + d];
            lsum += f * lsm[w * 16 + row];
        }
        const int hh = hk * G + rh;
        const float inv = (lsum == 0.f || lsum != lsum) ? 1.f : 1.f / lsum;
        if (num_splits == 1) {
            ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + d] = X::cvt(acc * inv);
            if (p.softmax_lse && d == 0)
                p.softmax_lse[(int64_t)b * p.h + hh] = (lsum == 0.f) ? INFINITY : (mx * p.softmax_scale + __logf(lsum));
        } else {
            float* oacc = (float*)p.workspace;
            float* lacc = oacc + (int64_t)num_splits * p.b * p.h * HD;
            const int64_t row_idx = ((int64_t)split * p.b + b) * p.h + hh;
            oacc[row_idx * HD + d] = acc * inv;
            if (d == 0) lacc[row_idx] = (lsum == 0.f) ? -INFINITY : (mxs + __log2f(lsum));   // log2 domain
        }
    }
}

// LSE-weighted merge of the split partials (flash_fwd_kernel.h:1116-1297). One 128-thread block per output row
// (b, q, h): the split weights are computed once (lanes over splits), then every thread owns one d and streams its
// partials with independent loads.  Serves the decode form (sq = 1) and the KV-split prefill form.
// workspace: float o_part[splits][b][sq][h][HD]; float lse_part[splits][b][sq][h]  (log2 domain)
template <typename T, int HD>
__global__ __launch_bounds__(128) void combine_kernel(vattn_attn_params p, int num_splits, int sq) {   // 128 threads: one per split weight, first HD also one per output column
    __shared__ float wsm[128];
    __shared__ float red[4];
    const int64_t row = blockIdx.x;                  // (b * sq + q) * h + head
    const int hh = (int)(row % p.h);
    const int64_t bq = row / p.h;
    const int q = (int)(bq % sq), b = (int)(bq / sq);
    const int tid = threadIdx.x;
    const float* oacc = (const float*)p.workspace;
    const int64_t sstride = (int64_t)p.b * sq * p.h;
    const float* lacc = oacc + (int64_t)num_splits * sstride * HD;
    const float my = (tid < num_splits) ? lacc[(int64_t)tid * sstride + row] : -INFINITY;    // num_splits <= 128
    float mx = my;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, xor_shuffle(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    const float w = (tid < num_splits) ? fast_exp2(my - mxs) : 0.f;
    float ws = w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ws += xor_shuffle(ws, o);
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = ws;
    wsm[tid] = w;
    __syncthreads();
    const float wsum = red[2] + red[3];
    const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
    if (tid < HD) {
        const float* src = oacc + row * HD + tid;
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < num_splits; s++) acc += wsm[s] * src[(int64_t)s * sstride * HD];
        ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)q * p.o_row_stride + (int64_t)hh * p.o_head_stride + tid] = Tr<T>::cvt(acc * inv);
    }
    if (p.softmax_lse && tid == 0)
        p.softmax_lse[((int64_t)b * p.h + hh) * sq + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
}

// Same merge for the KV-split prefill form, where there are b * sq * h output rows (tens of thousands) and at most 16
// partials each: one WAVE per row (4 rows per 256-thread block), lane l < splits holds partial l's LSE, the weights are
// broadcast by readlane, every lane owns two adjacent d.
template <typename T, int HD>
__global__ __launch_bounds__(256) void combine_rows_kernel(vattn_attn_params p, int num_splits, int sq, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (b * sq + q) * h + head
    if (row >= rows) return;
    const int hh = (int)(row % p.h);
    const int64_t bq = row / p.h;
    const int q = (int)(bq % sq), b = (int)(bq / sq);
    if (p.q_lens && q >= p.q_lens[b]) return;             // batched chunks: rows past this entry's length were never produced
    const int64_t q_first = p.q_start ? p.q_start[b] : 0;
    const float* oacc = (const float*)p.workspace;
    const int64_t sstride = rows;
    const float* lacc = oacc + (int64_t)num_splits * sstride * HD;
    const float my = (lane < num_splits) ? lacc[(int64_t)lane * sstride + row] : -INFINITY;
    float mx = my;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, xor_shuffle(mx, o));      // splits <= 16 live in lanes 0..15
    mx = __shfl(mx, 0, 64);
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    const float w = (lane < num_splits) ? fast_exp2(my - mxs) : 0.f;
    float wsum = w;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) wsum += xor_shuffle(wsum, o);
    wsum = __shfl(wsum, 0, 64);
    const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
    if (2 * lane < HD) {
        const float* src = oacc + row * HD + 2 * lane;
        float a0 = 0.f, a1 = 0.f;
        for (int s = 0; s < num_splits; s++) {
            const float ws = __shfl(w, s, 64);
            const float2 v = *(const float2*)(src + (int64_t)s * sstride * HD);
            a0 += ws * v.x;
            a1 += ws * v.y;
        }
        T* dst = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + q) * p.o_row_stride + (int64_t)hh * p.o_head_stride + 2 * lane;
        dst[0] = Tr<T>::cvt(a0 * inv);
        dst[1] = Tr<T>::cvt(a1 * inv);
    }
    if (p.softmax_lse && lane == 0)
        p.softmax_lse[((int64_t)b * p.h + hh) * sq + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
}

// ============================================================================================
// hardware-layout self test
// ============================================================================================

// Checks, against plain integer arithmetic, the three layout facts the kernels rely on:
//  [0] 32x32x16 C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
//  [1] 16x16x32 C/D map: col = lane&15, row = 4*(lane>>4) + r
//  [2] ds_read_b64_tr_b16: lane i of a 16-lane group receives element (i&3) of the 8-byte chunks
//      addressed by lanes 4*j + (i>>2), j = 0..3, of the same group
//  [3] A/B operands: lane (x = lane&31, g = lane>>5) contributes row/col x with k-slots (g, 0..7) (32x32x16)
//  [4] same for 16x16x32 with g = lane>>4
__global__ void selftest_kernel(int* res) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    const int lane = threadIdx.x;
    // [0],[3]: A = one-hot rows, B = one-hot cols with distinct values -> C[m][n] = sum_k A[m][k]B[k][n]
    {
        // A[m][slot] = (m + 1) if slot == (m & 15) else 0 ; B[slot][n] = (n + 1) * 64 + ... keep small ints
        f16x8 a, bq;
        const int x = lane & 31, g = lane >> 5;
        for (int j = 0; j < 8; j++) {
            const int slot = 8 * g + j;                 // logical k index shared by A and B
            a[j] = (_Float16)((slot == (x & 15)) ? (float)(x + 1) : 0.f);      // A[m=x][k]
            bq[j] = (_Float16)((slot == 3) ? 0.f : 0.f);
        }
        // B[k][n=x] = 1 for every k -> C[m][n] = sum_k A[m][k] = m + 1 for every n
        for (int j = 0; j < 8; j++) bq[j] = (_Float16)1.f;
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, c, 0, 0, 0);
        int bad = 0;
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            if (c[r] != (float)(row + 1)) bad = 1;
        }
        // columns: A[m][k] = 1 for all, B[k][n] = (n+1) if k-slot == (n & 15) -> C[m][n] = n + 1
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)1.f;
            bq[j] = (_Float16)(((8 * g + j) == (x & 15)) ? (float)(x + 1) : 0.f);
        }
        f32x16 c2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, c2, 0, 0, 0);
        int bad3 = 0;
        for (int r = 0; r < 16; r++)
            if (c2[r] != (float)(x + 1)) bad3 = 1;
        if (bad) atomicOr(&res[0], 1);
        if (bad3) atomicOr(&res[3], 1);
    }
    {
        f16x8 a, bq;
        const int x = lane & 15, g = lane >> 4;
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)(((8 * g + j) == x) ? (float)(x + 1) : 0.f);
            bq[j] = (_Float16)1.f;
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c, 0, 0, 0);
        int bad = 0;
        for (int r = 0; r < 4; r++)
            if (c[r] != (float)(4 * g + r + 1)) bad = 1;
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)1.f;
            bq[j] = (_Float16)(((8 * g + j) == x) ? (float)(x + 1) : 0.f);
        }
        f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c2, 0, 0, 0);
        int bad4 = 0;
        for (int r = 0; r < 4; r++)
            if (c2[r] != (float)(x + 1)) bad4 = 1;
        if (bad) atomicOr(&res[1], 1);
        if (bad4) atomicOr(&res[4], 1);
    }
    {
        // each lane owns the 8-byte chunk at lds[lane*4 .. lane*4+3]; value encodes (lane, element)
        for (int e = 0; e < 4; e++) lds[lane * 4 + e] = (short)(lane * 4 + e);
        __syncthreads();
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, &lds[lane * 4]));
        const int grp = lane >> 4, i = lane & 15;
        int bad = 0;
        for (int j = 0; j < 4; j++) {
            const int src_lane = grp * 16 + 4 * j + (i >> 2);
            if (t[j] != (short)(src_lane * 4 + (i & 3))) bad = 1;
        }
        if (bad) atomicOr(&res[2], 1);
    }
}

// ============================================================================================
// host side
// ============================================================================================

thread_local std::string g_err;
int fail(int code, const char* msg) {
    g_err = msg;
    return code;
}

// Split count for the decode form.  The kernel is built for 3 workgroups per CU (<= 168 VGPRs, 33 KiB LDS), i.e.
// 768 resident workgroups on 256 CUs; like the reference's heuristic (flash_api.cpp:258-323) pick the smallest
// split count whose last "round" of workgroups is nearly full, but against THIS chip's residency.
int pick_splits(const vattn_attn_params* p, int gblocks) {
    if (p->num_splits > 0) return p->num_splits > 128 ? 128 : p->num_splits;
    const long wg = (long)p->b * p->h_k * gblocks;
    const long slots = 768;
    const int max_len = p->seqlen_k + p->seqlen_knew;
    const int tiles = (max_len + DC_BN - 1) / DC_BN;
    long cap = tiles / 4;                       // at least one 32-key tile per wave and split
    if (cap < 1) cap = 1;
    // a split shorter than ~700 keys costs more in prologue / merge / combine than it returns: B1@32k 20.4 us at 32-48 splits
    // vs 26 us at 128; short contexts still want one tile per wave (B1@2k: 16 splits 11 us vs 24 us unsplit)
    if (cap > 48) cap = 48;
    if (wg * 10 >= slots * 6) return 1;      // the batch alone (nearly) fills the chip: splitting only adds combine work
    // otherwise: fill whole rounds of resident workgroups exactly (measured on MI355X, tools/kbench.py --splits:
    // 16 x 4 heads @32k: 12 splits = 768 workgroups 71.4 % of HBM peak vs 63.9-68.8 % for 4/6/8/16/24)
    double best = 0.0;
    long pick = 1;
    for (long s = 1; s <= cap; s++) {
        const double waves = (double)(wg * s) / slots;
        const double eff = waves / (double)((wg * s + slots - 1) / slots);
        if (eff > best + 1e-9) { best = eff; pick = s; }
    }
    return (int)pick;
}

void launch_append(const vattn_attn_params* p, hipStream_t st) {
    const int total = p->seqlen_knew * p->h_k * (p->d / 8);
    dim3 grid((total + 255) / 256, p->b), block(256);
    hipLaunchKernelGGL(append_kv_kernel, grid, block, 0, st, *p);
}

// variant bits 5-6: workgroup order (wg_to_work): 0 = default (XCD-grouped when the kv heads divide the 8 XCDs),
// 1 = block-major per head (3-D grid), 2 = heaviest-first across heads, 3 = XCD-grouped
dim3 prefill_grid(const vattn_attn_params* p, int nqb, int* order_out) {
    int order = (p->variant >> 5) & 3;
    order = order == 0 ? 2 : order - 1;
    if (order == 2 && !(p->h_k <= 8 && 8 % p->h_k == 0)) order = 1;
    dim3 grid(nqb, p->h, p->b);
    if (order == 1) grid = dim3((unsigned)(nqb * p->h * p->b));
    if (order == 2) {
        const int per = 8 / p->h_k;
        const long items = (long)nqb * p->b * (p->h / p->h_k);   // per kv head
        grid = dim3((unsigned)(8 * ((items + per - 1) / per)));
    }
    *order_out = order;
    return grid;
}

// Prefill plan: tiling and KV split, decided on the host from the shapes (used by the launch and by the workspace query).
//  tiling  0/1 = 8 waves x 32 rows, 2 = 4 waves x 64 rows, 4 = 4 waves x 32 rows, 6 = hand-interleaved 8-wave kernel.
//  nsplit  > 1 when the grid would leave CUs idle (tensor-parallel shards with few heads, short chunks): every work item's
//          key range is divided over nsplit workgroups, fp32 partials go through the workspace, combine_kernel merges them.
struct PrefillPlan { int tiling; int nsplit; };
PrefillPlan plan_prefill(const vattn_attn_params* p) {
    PrefillPlan pl;
    pl.tiling = (p->variant >> 1) & 7;
    pl.nsplit = 1;
    const bool auto_tiling = pl.tiling == 0;
    // 2 (64-row waves) and 6 (hand-interleaved, software-pipelined) exist for d = 128 only; 3 and 5 were the compiler-scheduled
    // pipelined and the phase-staggered kernels of round 1 (both slower, removed: profiles/r01_prefill_ablations.md)
    if (pl.tiling == 3 || pl.tiling == 5 || (p->d != 128 && (pl.tiling == 2 || pl.tiling == 6))) pl.tiling = 1;
    if (pl.tiling == 6 && p->q_lens) pl.tiling = 1;                  // the interleaved kernel has no batched-chunk form
    if (pl.tiling == 6) return pl;                                   // no split epilogue in that kernel
    // keys an average query block sees; without a host-side length only the chunk itself is certain
    const long lk = p->max_seqlen_k_hint > 0 ? p->max_seqlen_k_hint : p->seqlen_q;
    const long keys = p->is_causal ? (lk - p->seqlen_q / 2) : lk;
    const long tiles = keys > 0 ? (keys + PF_BN - 1) / PF_BN : 1;
    auto cap_by_tiles = [&](long want) {                             // >= 8 tiles (512 keys) per split: below that the
        if (want > 8) want = 8;                                      // partials cost more than they return
        while (want > 1 && tiles / want < 8) want--;
        return want < 1 ? 1 : (int)want;
    };
    // equal-length work items (a chunk on a long prefix): the split count whose last round of resident workgroups is
    // fullest, smallest such count if one is (nearly) exact; unequal lengths (causal whole prompt): two rounds, so that the
    // dispatcher's heaviest-first order can even them out
    const bool uniform = !p->is_causal || lk >= 4L * p->seqlen_q;
    auto pick = [&](long wg, long slots) {
        const int cap = cap_by_tiles(8);
        if (!uniform) return cap_by_tiles((2 * slots + wg - 1) / wg);
        int best = 1;
        double best_eff = 0.0;
        for (int ns = 1; ns <= cap; ns++) {
            const double rounds = (double)(wg * ns) / slots;
            const double eff = rounds / (double)((wg * ns + slots - 1) / slots);
            if (eff >= 0.95) return ns;
            if (eff >= best_eff - 1e-9) { best_eff = eff; best = ns; }
        }
        return best;
    };
    const long wg8 = (long)((p->seqlen_q + 255) / 256) * p->h * p->b;     // 8-wave workgroups: one per CU
    const long wg4 = (long)((p->seqlen_q + 127) / 128) * p->h * p->b;     // 4-wave workgroups: two per CU
    if (p->num_splits > 0) {                                         // forced (tests, benchmarks)
        if (auto_tiling && wg8 <= 256) pl.tiling = 4;
        pl.nsplit = p->num_splits > 16 ? 16 : p->num_splits;
        return pl;
    }
    if (!auto_tiling) {                                              // explicit tiling: split only an underfilled grid
        const long slots = pl.tiling == 4 ? 512 : 256, wg = pl.tiling == 4 ? wg4 : wg8;
        if (wg < slots) pl.nsplit = pick(wg, slots);
        return pl;
    }
    // Default.  A grid of >= 256 eight-wave workgroups fills the chip: no split (above one workgroup per CU the 8-wave
    // tiling wins by 1-7 %; at exactly one per CU, causal work of very unequal length, the 4-wave tiling measures +19-22 %:
    // Llama-70B/TP8 8k prompt 552 -> 676 TFLOP/s, 2k prompt 500 -> 594).  Below that (tensor-parallel shards with few heads,
    // short chunks on long prefixes) split the key range over 8-wave workgroups; if even 8 splits leave most CUs idle, take
    // the 4-wave tiling.  Measured (tools/kbench.py --pf-splits, profiles/r01_kbench.txt): Llama-70B/TP8 2k chunk @ 30k
    // 393 -> 901 TFLOP/s, 512 chunk @ 16k 105 -> 568, Yi-34B/TP2 1k chunk @ 64k 653 -> 850.
    if (wg8 > 256) return pl;
    if (wg8 == 256) { pl.tiling = 4; return pl; }
    const int ns8 = pick(wg8, 256);
    if (ns8 == 1) { pl.tiling = 4; return pl; }                      // cannot split (short prefix): more, smaller workgroups
    if (wg8 * ns8 >= 192) { pl.nsplit = ns8; return pl; }
    pl.tiling = 4;
    pl.nsplit = pick(wg4, 512);
    return pl;
}

template <typename T, int HD, int WAVES, int QC, bool MSUM> void launch_prefill(const vattn_attn_params* p, hipStream_t st, bool use_tr, int nsplit) {
    constexpr int BM = 32 * QC * WAVES;
    const int nqb = (p->seqlen_q + BM - 1) / BM;
    int order;
    dim3 grid = prefill_grid(p, nqb, &order);
    const dim3 block(64 * WAVES);
    if (nsplit > 1) {
        if (order == 0) {      // the split lives in the 1-D orders
            vattn_attn_params q = *p;
            q.variant = (p->variant & ~(3 << 5)) | (2 << 5);
            grid = prefill_grid(&q, nqb, &order);
        }
        grid = dim3(((grid.x + 7) / 8) * 8 * nsplit);
    }
    const size_t smem = PfSmem<HD>::kTotal;
    static const bool attr_once = [] {   // 64 KiB of dynamic LDS per workgroup
        (void)hipFuncSetAttribute((const void*)prefill_kernel<T, HD, true, WAVES, QC, MSUM>, hipFuncAttributeMaxDynamicSharedMemorySize, PfSmem<HD>::kTotal);
        (void)hipFuncSetAttribute((const void*)prefill_kernel<T, HD, false, WAVES, QC, MSUM>, hipFuncAttributeMaxDynamicSharedMemorySize, PfSmem<HD>::kTotal);
        return true;
    }();
    (void)attr_once;
    if (use_tr)
        hipLaunchKernelGGL((prefill_kernel<T, HD, true, WAVES, QC, MSUM>), grid, block, smem, st, *p, order, nqb, nsplit);
    else
        hipLaunchKernelGGL((prefill_kernel<T, HD, false, WAVES, QC, MSUM>), grid, block, smem, st, *p, order, nqb, nsplit);
    if (nsplit > 1) {
        const int64_t rows = (int64_t)p->b * p->seqlen_q * p->h;
        hipLaunchKernelGGL((combine_rows_kernel<T, HD>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, *p, nsplit, p->seqlen_q, rows);
    }
}

template <typename T, int HD> int launch_attn_t(const vattn_attn_params* p, hipStream_t st, bool time_only_main) {
    (void)time_only_main;
    const bool use_tr = (p->variant & 1) == 0;
    if (p->seqlen_q == 1) {
        const int G = p->h / p->h_k;
        const int gblocks = (G + 15) / 16;
        const int splits = pick_splits(p, gblocks);
        if (splits > 1 && !p->workspace) return fail(VATTN_K_ERR_INVALID, "split-KV decode needs a workspace");
        dim3 grid(splits, p->h_k * gblocks, p->b), block(64 * DC_WAVES);
        const size_t smem = (size_t)DC_WAVES * 16 * HD * 4 + DC_WAVES * 16 * 4 * 2;   // merge area >= V staging (4*8 KiB)
        const int fused_append = (p->k_new && p->seqlen_knew == 1) ? 1 : 0;
        if (p->k_new && !fused_append) launch_append(p, st);        // seqlen_knew > 1: separate append launch
        if (use_tr)
            hipLaunchKernelGGL((decode_kernel<T, HD, true>), grid, block, smem, st, *p, splits, gblocks, fused_append);
        else
            hipLaunchKernelGGL((decode_kernel<T, HD, false>), grid, block, smem, st, *p, splits, gblocks, fused_append);
        if (splits > 1) hipLaunchKernelGGL((combine_kernel<T, HD>), dim3(p->b * p->h), dim3(128), 0, st, *p, splits, 1);
    } else {
        if (p->k_new && p->seqlen_knew > 0) launch_append(p, st);
        const PrefillPlan pl = plan_prefill(p);
        if (pl.nsplit > 1 && !p->workspace) return fail(VATTN_K_ERR_INVALID, "KV-split prefill needs a workspace (vattn_attn_workspace_bytes)");
        bool launched = false;
        if constexpr (HD == 128) {
            if (pl.tiling == 2) {
                launch_prefill<T, 128, 4, 2, false>(p, st, use_tr, pl.nsplit);
                launched = true;
            } else if (pl.tiling == 6) {
                const int nqb = (p->seqlen_q + 255) / 256;
                static const bool once6 = [] {
                    (void)hipFuncSetAttribute((const void*)prefill_ilv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, PfSmem<128>::kTotal);
                    return true;
                }();
                (void)once6;
                int order;
                const dim3 grid = prefill_grid(p, nqb, &order);
                hipLaunchKernelGGL((prefill_ilv_kernel<T>), grid, dim3(512), PfSmem<128>::kTotal, st, *p, order, nqb);
                launched = true;
            }
        }
        if (launched) {
        } else if (pl.tiling == 4) launch_prefill<T, HD, 4, 1, false>(p, st, use_tr, pl.nsplit);
        else if ((p->variant & 16) && HD == 128) launch_prefill<T, HD == 128 ? 128 : HD, 8, 1, HD == 128>(p, st, use_tr, pl.nsplit);      // denominator on the matrix pipe
        else launch_prefill<T, HD, 8, 1, false>(p, st, use_tr, pl.nsplit);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

int validate(const vattn_attn_params* p) {
    if (!p || !p->q || !p->out || !p->k_cache || !p->v_cache) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    if (p->dtype != VATTN_DTYPE_F16 && p->dtype != VATTN_DTYPE_BF16)
        return fail(VATTN_K_ERR_UNSUPPORTED, "FlashAttention only support fp16 and bf16 data type");      // flash_api.cpp:1325-1326
    // d = 256 instantiates but spills (O^T alone is 128 accumulator registers per wave): not shipped until it has its own tiling
    if (p->d != 64 && p->d != 128) return fail(VATTN_K_ERR_UNSUPPORTED, "this build supports head dimensions 64 and 128");
    if (p->b <= 0) return fail(VATTN_K_ERR_INVALID, "batch size must be postive");                       // flash_api.cpp:1353
    if (p->h_k <= 0 || p->h % p->h_k != 0)
        return fail(VATTN_K_ERR_INVALID, "Number of heads in key/value must divide number of heads in query");   // :1355
    if ((p->k_new == nullptr) != (p->v_new == nullptr))
        return fail(VATTN_K_ERR_INVALID, "If key is supplied, value must also be passed in");            // :1452
    if (p->k_new && !p->cache_seqlens)
        return fail(VATTN_K_ERR_INVALID, "If key is supplied, seqlens_k must also be passed in");        // :1453
    if (p->seqlen_q <= 0 || p->seqlen_k < 0) return fail(VATTN_K_ERR_INVALID, "bad sequence lengths");
    if ((p->q_start == nullptr) != (p->q_lens == nullptr)) return fail(VATTN_K_ERR_INVALID, "q_start and q_lens must be given together");
    if (p->q_lens && p->seqlen_q == 1) return fail(VATTN_K_ERR_UNSUPPORTED, "batched chunks need max(q_lens) > 1 (the decode form is already batched)");
    if (p->q_lens && p->k_new) return fail(VATTN_K_ERR_UNSUPPORTED, "batched chunks: append the new keys/values with cache_flat first");
    // 16-byte vector access requirements
    const int64_t strides[] = {p->q_batch_stride, p->q_row_stride, p->q_head_stride, p->k_batch_stride, p->k_row_stride,
                               p->k_head_stride, p->v_batch_stride, p->v_row_stride, p->v_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return fail(VATTN_K_ERR_UNSUPPORTED, "strides must be multiples of 8 elements (16-byte vector access)");
    if (((uintptr_t)p->q | (uintptr_t)p->k_cache | (uintptr_t)p->v_cache | (uintptr_t)p->out) & 15)
        return fail(VATTN_K_ERR_UNSUPPORTED, "tensor base pointers must be 16-byte aligned");
    if (p->o_row_stride % 4 != 0 || p->o_head_stride % 4 != 0 || p->o_batch_stride % 4 != 0)
        return fail(VATTN_K_ERR_UNSUPPORTED, "output strides must be multiples of 4 elements");
    return VATTN_K_OK;
}

}  // namespace

extern "C" {

const char* vattn_kernels_last_error(void) { return g_err.c_str(); }

size_t vattn_attn_workspace_bytes(const vattn_attn_params* p) {
    if (!p || p->h_k <= 0 || p->h <= 0 || p->b <= 0 || p->seqlen_q <= 0) return 0;
    if (p->seqlen_q != 1) {
        const int ns = plan_prefill(p).nsplit;
        return ns > 1 ? (size_t)ns * p->b * p->seqlen_q * p->h * (p->d + 1) * sizeof(float) : 0;
    }
    const int G = p->h / p->h_k;
    const int gblocks = (G + 15) / 16;
    const int splits = pick_splits(p, gblocks);
    if (splits <= 1) return 0;
    return (size_t)splits * p->b * p->h * (p->d + 1) * sizeof(float);
}

int vattn_flash_attn_with_kvcache(const vattn_attn_params* p, void* stream) {
    int rc = validate(p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (p->k_new && p->seqlen_knew > 0 && !p->cache_seqlens) return fail(VATTN_K_ERR_INVALID, "If key is supplied, seqlens_k must also be passed in");
    const bool f16 = p->dtype == VATTN_DTYPE_F16;
    switch (p->d) {
        case 64: return f16 ? launch_attn_t<_Float16, 64>(p, st, false) : launch_attn_t<__bf16, 64>(p, st, false);
        default: return f16 ? launch_attn_t<_Float16, 128>(p, st, false) : launch_attn_t<__bf16, 128>(p, st, false);
    }
}

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

int vattn_selftest_layouts(void* stream, int32_t* detail_out) {
    hipStream_t st = (hipStream_t)stream;
    int* d = nullptr;
    if (hipMalloc(&d, 8 * sizeof(int)) != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, "hipMalloc failed");
    hipMemsetAsync(d, 0, 8 * sizeof(int), st);
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, d);
    int h[8] = {0};
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    int bad = 0;
    for (int i = 0; i < 8; i++) {
        if (detail_out) detail_out[i] = h[i];
        bad |= h[i];
    }
    return bad ? fail(VATTN_K_ERR_INVALID, "hardware layout assumption violated") : VATTN_K_OK;
}

float vattn_time_attn(const vattn_attn_params* p, void* stream, int32_t warmup, int32_t iters) {
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < warmup; i++)
        if (vattn_flash_attn_with_kvcache(p, stream) != 0) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; i++)
        if (vattn_flash_attn_with_kvcache(p, stream) != 0) return -1.f;
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / (iters > 0 ? iters : 1);
}

}  // extern "C"
