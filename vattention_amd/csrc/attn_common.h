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

// Two builds of these sources.  The PRODUCT library (libvattn_amd.so) instantiates the kernels the launch plans choose, and accepts
// only the `variant` bits that select between them (explicit tiling / workgroup order / workgroup shape: what a benchmark needs to
// A/B the plan against its alternatives).  The LAB library (-DVATTN_LAB, tools/lab/libvattn_lab.so, built by vattention_amd/build.py
// for tools/kbench.py and the variant tests) adds the measurement scaffolding: alternative operand paths and schedules, the
// in-launch merge protocols that measured slower than two launches, and timing ablations whose RESULTS ARE WRONG.
#ifdef VATTN_LAB
constexpr bool kLab = true;
#else
constexpr bool kLab = false;
#endif
// variant bits of vattn_attn_params a product build accepts: bits 1-3 tiling {0 plan, 1 = 8 waves x 32 rows, 4 = 4 waves x 32 rows,
// 7 = prefill64}; bits 5-6 workgroup order; bit 7 one 16-head block per decode workgroup; bits 12-13 role policy of the fused launch;
// bit 19: decode keeps the grid heuristics of rounds 1-3 (uniform split / host plan) instead of the device-planned stream decomposition
// (an A/B selector).  LAB ONLY, bit 20: the stream decomposition merges inside the decode launch (tickets) instead of in a second launch.
constexpr int kVariantLegacyDecodePlan = 1 << 19, kVariantInLaunchMerge = 1 << 20;
constexpr int kProductVariantMask = (7 << 1) | (3 << 5) | (1 << 7) | (3 << 12) | kVariantLegacyDecodePlan;

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

// max over the two half-waves, in every lane: both outputs of the swap already hold (lo, lo) and (hi, hi) — no select, and an asm
// v_max so that the compiler does not canonicalise the operands first (two extra v_max x, x per call on MFMA outputs)
__device__ __forceinline__ float max_halves(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
    return m;
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


// ---- device-scope (cache-bypassing) accesses for data handed from one workgroup to another inside a launch ----
// Relaxed atomics at agent scope compile to global_store/load ... sc1: the store is written through to the device's coherence
// point and the load does not hit a stale line of this XCD's L2 — no fence (= no L2 write-back + invalidate) needed around them.
// The same for buffer accesses: cache-policy operand of the raw buffer intrinsics with the SC1 bit (device scope on gfx940+): a store is
// written through, a load is not served from this CU's L1.  16-byte device-scope stores cost what plain ones do.
constexpr int kDevScope = 16;
__device__ __forceinline__ void store_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- single-launch merge of KV-split prefill partials (round 2) ----
// Called by every workgroup of a KV-split prefill launch after it has written its partial (normalised fp32 rows + log2-domain LSE,
// layout of combine_rows_kernel): release (barrier, then ONE wave's agent-scope fence), ticket from the query block's counter, and
// the holder of the last ticket merges the nsplit partials of the block's rows [q0, q_end) of head h — the arithmetic of
// combine_rows_kernel (one wave per row, lanes over column pairs), without the second launch and with the partials still in L2 /
// MALL.  `counter` is zero between launches (the merger resets it).
// mode 1: partials written with ordinary stores, ordered by agent-scope fences (each writes back and invalidates the XCD's L2);
// mode 2: partials written and read with device-scope accesses (store_dev / load_dev), ordered by the barrier's vmcnt(0) alone.
template <typename T, int HD>
__device__ __forceinline__ void prefill_release_and_merge(const vattn_attn_params& p, const int nsplit, const int b, const int h, const int q0,
                                                          const int q_end, const int64_t q_first, int* counter, int* s_ticket, const int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    __syncthreads();                                   // every wave's partial stores have completed (s_waitcnt vmcnt(0) + barrier)
    if (tid < 64) {
        if (mode == 1) __threadfence();
        if (tid == 0) *s_ticket = atomicAdd(counter, 1);
    }
    __syncthreads();
    if (*s_ticket != nsplit - 1) return;
    if (mode == 1 && tid < 64) __threadfence();
    __syncthreads();
    const bool dev = mode == 2;
    const int sq = p.seqlen_q;
    const float* oacc = (const float*)p.workspace;
    const int64_t sstride = (int64_t)p.b * sq * p.h;
    const float* lacc = oacc + (int64_t)nsplit * sstride * HD;
    for (int q = q0 + wave; q < q_end; q += nwaves) {
        const int64_t row = ((int64_t)b * sq + q) * p.h + h;
        float my = -INFINITY;
        if (lane < nsplit) my = dev ? load_dev(lacc + (int64_t)lane * sstride + row) : lacc[(int64_t)lane * sstride + row];
        float mx = my;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, xor_shuffle(mx, o));      // splits <= 16 live in lanes 0..15
        mx = __shfl(mx, 0, 64);
        const float mxs = (mx == -INFINITY) ? 0.f : mx;
        const float w = (lane < nsplit) ? fast_exp2(my - mxs) : 0.f;
        float wsum = w;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) wsum += xor_shuffle(wsum, o);
        wsum = __shfl(wsum, 0, 64);
        const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
        if (2 * lane < HD) {
            const float* src = oacc + row * HD + 2 * lane;
            float a0 = 0.f, a1 = 0.f;
            for (int s = 0; s < nsplit; s++) {
                const float ws = __shfl(w, s, 64);
                const float* sp = src + (int64_t)s * sstride * HD;
                float2 v;
                if (dev) { v.x = load_dev(sp); v.y = load_dev(sp + 1); }
                else v = *(const float2*)sp;
                a0 += ws * v.x;
                a1 += ws * v.y;
            }
            T* dst = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + q) * p.o_row_stride + (int64_t)h * p.o_head_stride + 2 * lane;
            dst[0] = Tr<T>::cvt(a0 * inv);
            dst[1] = Tr<T>::cvt(a1 * inv);
        }
        if (p.softmax_lse && lane == 0)
            p.softmax_lse[((int64_t)b * p.h + h) * sq + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
    }
    if (tid == 0) *counter = 0;
}

// ---- host-side pieces shared by the translation units ----
int fail(int code, const char* msg);                                    // attn_api.hip: records the message for vattn_kernels_last_error
void launch_append(const vattn_attn_params* p, hipStream_t st);         // cache_kernels.hip
int launch_prefill_form(const vattn_attn_params* p, hipStream_t st);    // prefill_kernels.hip (seqlen_q > 1)
size_t prefill_workspace_bytes(const vattn_attn_params* p);
int prefill_worklist(const vattn_attn_params* p, const int32_t* q_lens, const int32_t* k_lens, vattn_prefill_item* items, int cap_items,
                     vattn_prefill_item* blocks, int cap_blocks, int32_t* counts, int32_t* wg_first = nullptr, int max_wg = 0, int persist_mode = 0);   // prefill_kernels.hip; persist_mode 0 none, 1 host-assigned queues, 2 drawn queues
void launch_prefill64p(const vattn_attn_params* p, hipStream_t st, int* ctr);      // prefill64p_kernels.hip: persistent workgroups over a work list (ctr: drawn queues)
int* queue_counters(hipStream_t st);      // attn_api.hip: 8 zeroed ints per (device, stream) for the drawn queues, NULL while the stream is being captured before they exist
void launch_prefill64(const vattn_attn_params* p, hipStream_t st, int nsplit);      // prefill64_kernels.hip (d = 128)
#ifdef VATTN_LAB
void launch_prefill64(const vattn_attn_params* p, hipStream_t st, int nsplit, int* done, int merge_mode);   // tools/lab/csrc/prefill64_lab.hip; done: counters of the single-launch merge or NULL
#endif
int* merge_counters(hipStream_t st, size_t n_ints);                      // attn_api.hip: zeroed per-(device, stream) counters, NULL while capturing
int launch_decode_form(const vattn_attn_params* p, hipStream_t st);     // decode_kernels.hip (seqlen_q == 1)
size_t decode_workspace_bytes(const vattn_attn_params* p);
int decode_plan(const vattn_attn_params* p, const int32_t* lens, vattn_decode_item* items, int cap, int32_t* seq);   // decode_kernels.hip
void prefill_describe(const vattn_attn_params* p, vattn_plan_desc* out);   // prefill_kernels.hip
void decode_describe(const vattn_attn_params* p, vattn_plan_desc* out);    // decode_kernels.hip
int launch_hybrid(const vattn_attn_params* prefill, const vattn_attn_params* decode, void* ws, hipStream_t st);   // hybrid_kernels.hip
size_t hybrid_workspace_bytes(const vattn_attn_params* prefill, const vattn_attn_params* decode);

}  // namespace vattn_k
