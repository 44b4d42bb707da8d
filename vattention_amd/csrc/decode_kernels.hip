// Decode form (seqlen_q == 1) of flash_attn_with_kvcache on gfx950: GQA group packed into the MFMA N dimension, split-KV over
// the context, K fragments loaded straight from HBM into MFMA operand registers, V through a wave-private LDS transpose
// stage, fp32 online softmax, in-workgroup merge of the 4 waves, new K/V row appended in-kernel, LSE-weighted combine across
// splits (combine_kernel; flash_fwd_kernel.h:1116-1297).  Call-site semantics: flash_api.cpp:1367-1378,1451-1454,1558-1560.
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// decode (seqlen_q == 1): split-KV
// ============================================================================================

}  // namespace vattn_k
#include "decode_body.h"  // DC_WAVES, DC_BN, decode_body / decode_kernel
namespace vattn_k {


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
    // up to kEarly partials per thread are asked for BEFORE the weights exist: their loads fly together with the LSE loads instead of
    // behind the two reductions (the kernel is two dependent memory latencies long, nothing else)
    constexpr int kEarly = 16;
    const bool early = num_splits <= kEarly;
    const float* src = oacc + row * HD + (tid < HD ? tid : 0);
    float part[kEarly];
    if (early) {
#pragma unroll
        for (int s = 0; s < kEarly; s++) part[s] = (s < num_splits && tid < HD) ? src[(int64_t)s * sstride * HD] : 0.f;
    }
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
        float acc = 0.f;
        if (early) {
#pragma unroll
            for (int s = 0; s < kEarly; s++) acc += wsm[s] * part[s];      // wsm[s] = 0 and part[s] = 0 beyond num_splits: the same sum, the same order
        } else {
#pragma unroll 8
            for (int s = 0; s < num_splits; s++) acc += wsm[s] * src[(int64_t)s * sstride * HD];
        }
        ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)q * p.o_row_stride + (int64_t)hh * p.o_head_stride + tid] = Tr<T>::cvt(acc * inv);
    }
    if (p.softmax_lse && tid == 0)
        p.softmax_lse[((int64_t)b * p.h + hh) * sq + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
}


// The same merge for a length-balanced plan (vattn_decode_plan): sequence b owns items [first, first + count) — count differs per
// sequence — and item i's partial rows are (i * h + head).
template <typename T, int HD>
__global__ __launch_bounds__(128) void combine_items_kernel(vattn_attn_params p) {
    __shared__ float wsm[128];
    __shared__ float red[4];
    const int64_t row = blockIdx.x;                  // b * h + head
    const int hh = (int)(row % p.h);
    const int b = (int)(row / p.h);
    const int tid = threadIdx.x;
    const int first = p.split_seq[2 * b], cnt = p.split_seq[2 * b + 1];      // cnt <= 128
    const float* oacc = (const float*)p.workspace;
    const float* lacc = oacc + (int64_t)p.num_split_items * p.h * HD;
    const float my = (tid < cnt) ? lacc[(int64_t)(first + tid) * p.h + hh] : -INFINITY;
    float mx = my;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, xor_shuffle(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    const float w = (tid < cnt) ? fast_exp2(my - mxs) : 0.f;
    float ws = w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ws += xor_shuffle(ws, o);
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = ws;
    wsm[tid] = w;
    __syncthreads();
    const float wsum = red[2] + red[3];
    const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
    if (tid < HD) {
        const float* src = oacc + ((int64_t)first * p.h + hh) * HD + tid;
        float acc = 0.f;
#pragma unroll 4
        for (int s = 0; s < cnt; s++) acc += wsm[s] * src[(int64_t)s * p.h * HD];
        ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + tid] = Tr<T>::cvt(acc * inv);
    }
    if (p.softmax_lse && tid == 0)
        p.softmax_lse[(int64_t)b * p.h + hh] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
}

// Split count for the decode form.  The kernel is built for 3 workgroups per CU (<= 168 VGPRs, 33 KiB LDS), i.e.
// 768 resident workgroups on 256 CUs; like the reference's heuristic (flash_api.cpp:258-323) pick the smallest
// split count whose last "round" of workgroups is nearly full, but against THIS chip's residency.
int pick_splits(const vattn_attn_params* p, int gblocks, long slots = 768) {
    if (p->num_splits > 0) return p->num_splits > 128 ? 128 : p->num_splits;
    const long wg = (long)p->b * p->h_k * gblocks;
    const int max_len = p->seqlen_k + p->seqlen_knew;
    const int tiles = (max_len + DC_BN - 1) / DC_BN;
    long cap = tiles / 4;                       // at least one 32-key tile per wave and split
    if (cap < 1) cap = 1;
    // a split shorter than ~700 keys costs more in prologue / merge / combine than it returns: B1@32k 20.4 us at 32-48 splits
    // vs 26 us at 128; short contexts still want one tile per wave (B1@2k: 16 splits 11 us vs 24 us unsplit)
    // -> at most 48 splits; 64 when fewer than four (sequence, kv head) groups exist and the context is long enough to keep ~700
    // keys per split (one 128 k sequence on a TP4 shard, 2 kv heads: 35.4 us at 48 splits, 30.8 at 64, 33.7 at 128; with 4 kv heads
    // 48 / 64 / 128 splits measure 52.3 / 52.8 / 53.4 us: profiles/r02_kbench_decode_splits.txt)
    // Round 4 re-measured these with the launches ROTATING over enough caches that the 256 MiB Infinity Cache cannot serve a repeated
    // launch (tools/kbench.py --rotate, profiles/r04_decode_b1_rotating_caches.txt: one 128 k sequence is 268 MB of K/V on a TP2 shard —
    // repeated on ONE cache it reads 51 us, over rotating caches 64 us, which is what a 60-layer model sees): 2 kv heads 47.0 / 40.5 /
    // 36.0 / 37.3 us at 48 / 64 / 96 / 128 splits, 4 kv heads 64.8 / 61.1 / 64.9 / 62.1 us, one 32 k sequence on 4 kv heads 24.8 / 26.1 / 27.8
    const long cap_len = (tiles / 21 < 64) ? 48 : (wg <= 2 ? 96 : wg <= 4 ? 64 : 48);
    if (cap > cap_len) cap = cap_len;
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

// Head blocks per workgroup: two when the kv head serves more than 16 query heads (one pass over K/V for 32 heads).
static inline int decode_nb(const vattn_attn_params* p) { return (p->h / p->h_k > 16 && !(p->variant & 128)) ? 2 : 1; }
static inline int decode_groups(const vattn_attn_params* p) {
    const int blocks = (p->h / p->h_k + 15) / 16;
    const int nb = decode_nb(p);
    return (blocks + nb - 1) / nb;
}
// (the alternative workgroup shapes — 8 / 16 waves, two K/V register sets per wave — and the single-launch merges measured slower and live
// in the lab copy, tools/lab/csrc/decode_kernels_lab.hip; profiles/r03_kbench_decode_shapes.txt, r02_kbench_decode_merge.txt)
static inline long decode_slots(const vattn_attn_params* p) {      // resident workgroups
    return (p->d == 128 && decode_nb(p) == 2) ? 512 : 768;
}
// Device-planned stream decomposition (decode_body.h, decode_stream_kernel): the host only picks the number of workgroups per (kv head,
// head-block group) — from the batch size and the cache VIEW's row count, which the reference's wrapper makes the batch's longest
// context (vattention_flashattention_wrapper.py:196-197: `kv_cache[0][:, :self.max_cache_len]`) — the lengths stay on the device.
// One round of the resident workgroups; fewer when the batch is small or its contexts short (see per_seq below).  0 = take the grid
// heuristics of rounds 1-3: ONE sequence (nothing to balance: its uniform split is the optimum and needs no plan prologue), GQA groups
// wider than 16 heads, batches beyond DC_MAXB, explicit num_splits > 0, variant bit 19.  num_splits = -N forces N workgroups per group (tests, A/B).
int stream_nwg(const vattn_attn_params* p) {
    if (p->seqlen_q != 1 || p->split_items || p->num_splits > 0 || (p->variant & kVariantLegacyDecodePlan)) return 0;
    if (p->b > DC_MAXB || decode_groups(p) != 1) return 0;
    const long slots = decode_nb(p) == 2 ? 512 : 768;      // resident workgroups of decode_stream_kernel (its launch bounds)
    const long gps = p->h_k;
    if (p->num_splits < 0) return (int)std::min<long>(-(long)p->num_splits, 65535);
    // ONE sequence has nothing to balance, and the two-block workgroups of wide GQA groups (16 < G <= 32) measure 16 % slower on this path
    // (mqa G32 B16 @ 16 k: 42.5 vs 36.5 us): both keep the grid heuristics
    if (p->b < 2 || decode_nb(p) == 2) return 0;
    const long max_tiles = std::max(1L, ((long)p->seqlen_k + DC_BN - 1) / DC_BN);      // (seqlen_k rows already hold the appended token)
    // pieces per sequence, on average: at least one tile per wave and piece, at most 48 (pick_splits' measurements: a piece shorter than
    // ~700 keys costs more in prologue and merge than it returns once the chip is full, short contexts still want every CU busy)
    const long per_seq = std::min(48L, std::max(1L, max_tiles / 4));
    const long nwg = std::min(std::max(1L, slots / gps), (long)p->b * per_seq);
    return (int)nwg;
}
static size_t stream_workspace_bytes(const vattn_attn_params* p, int nwg) {
    const size_t rf = decode_nb(p) == 2 ? (size_t)(32 * p->d + 32) : (size_t)(16 * p->d + 32);
    return stream_table_bytes(p->b) + (size_t)(nwg + p->b) * p->h_k * rf * sizeof(float);      // (first record, count) per sequence, then the records
}

template <typename T, int HD, int NB> int launch_decode_stream(const vattn_attn_params* p, hipStream_t st, int nwg) {
    if (!p->workspace) return fail(VATTN_K_ERR_INVALID, "split-KV decode needs a workspace");
    if (stream_workspace_bytes(p, nwg) >= 0x7fffffffull) return fail(VATTN_K_ERR_UNSUPPORTED, "decode batch too large for the 32-bit record offsets");
    const size_t smem = (size_t)DC_WAVES * 16 * HD * 4 + DC_WAVES * 16 * 4 * 2;
    const int fused_append = (p->k_new && p->seqlen_knew == 1) ? 1 : 0;
    if (p->k_new && !fused_append) launch_append(p, st);
    const dim3 grid((unsigned)nwg, (unsigned)p->h_k), block(64 * DC_WAVES);
    if constexpr (__is_same(T, __bf16)) {
        // bf16 rotates through fp32 (no packed arithmetic): with the fused-RoPE path compiled in, decode_stream_kernel<bf16, 128, one head block> is
        // 12 registers over the 168 of three workgroups per CU and gets a scratch segment — 9 us per launch even when no rotation is asked for
        // (profiles/r06_decode_bf16_scratch.txt).  Two builds: without the path (what the reference's wrapper calls: no spill), and the one that
        // takes it at run time (the path compiled in UNCONDITIONALLY spills more: 46 registers instead of 12).
        if (p->rotary_cos_sin) hipLaunchKernelGGL((decode_stream_kernel<T, HD, true, NB, -1>), grid, block, smem, st, *p, 1, fused_append);
        else hipLaunchKernelGGL((decode_stream_kernel<T, HD, true, NB, 0>), grid, block, smem, st, *p, 1, fused_append);
    } else hipLaunchKernelGGL((decode_stream_kernel<T, HD, true, NB>), grid, block, smem, st, *p, 1, fused_append);
    hipLaunchKernelGGL((decode_stream_combine_kernel<T, HD, NB>), dim3((unsigned)p->b, (unsigned)p->h_k), dim3(256), 0, st, *p, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

template <typename T, int HD, int NB> int launch_decode_nb(const vattn_attn_params* p, hipStream_t st) {
    constexpr int W = DC_WAVES;
    if (const int nwg = stream_nwg(p)) return launch_decode_stream<T, HD, NB>(p, st, nwg);
    const int groups = decode_groups(p);
    const bool planned = p->split_items != nullptr;
    if (planned && (!p->split_seq || p->num_split_items <= 0)) return fail(VATTN_K_ERR_INVALID, "split_items needs split_seq and num_split_items");
    const int splits = planned ? 2 : pick_splits(p, groups, decode_slots(p));
    if (splits > 1 && !p->workspace) return fail(VATTN_K_ERR_INVALID, "split-KV decode needs a workspace");
    dim3 grid(splits, p->h_k * groups, p->b), block(64 * W);
    if (planned) grid = dim3((unsigned)p->num_split_items, p->h_k * groups, 1);
    if (groups > 1 && !(p->variant & 64) && !planned) {            // sibling groups share an XCD (variant bit 6: plain 3-D grid, for A/B)
        const long w = (long)splits * p->h_k * p->b;
        grid = dim3((unsigned)(((w + 7) / 8) * 8 * groups));
    }
    const size_t smem = (size_t)W * 16 * HD * 4 + W * 16 * 4 * 2;   // merge area >= V staging (W x 8 KiB)
    const int fused_append = (p->k_new && p->seqlen_knew == 1) ? 1 : 0;
    if (p->k_new && !fused_append) launch_append(p, st);        // seqlen_knew > 1: separate append launch
    const vattn_attn_params& q = *p;
    hipLaunchKernelGGL((decode_kernel<T, HD, true, NB>), grid, block, smem, st, q, splits, groups, fused_append);
    if (planned) hipLaunchKernelGGL((combine_items_kernel<T, HD>), dim3(p->b * p->h), dim3(128), 0, st, q);
    else if (splits > 1) hipLaunchKernelGGL((combine_kernel<T, HD>), dim3(p->b * p->h), dim3(128), 0, st, q, splits, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

template <typename T, int HD> int launch_decode_t(const vattn_attn_params* p, hipStream_t st) {
    return decode_nb(p) == 2 ? launch_decode_nb<T, HD, 2>(p, st) : launch_decode_nb<T, HD, 1>(p, st);
}

int launch_decode_form(const vattn_attn_params* p, hipStream_t st) {
    const bool f16 = p->dtype == VATTN_DTYPE_F16;
    if (p->d == 64) return f16 ? launch_decode_t<_Float16, 64>(p, st) : launch_decode_t<__bf16, 64>(p, st);
    return f16 ? launch_decode_t<_Float16, 128>(p, st) : launch_decode_t<__bf16, 128>(p, st);
}

// Length-balanced split of a ragged decode batch (include/vattn_kernels.h, vattn_decode_plan).  Every sequence is cut into pieces of at
// most T tiles; T is the smallest piece length for which the pieces of ALL sequences fit the resident workgroups in one round
// (slots / (kv heads x head-block groups) items), or — when the batch alone exceeds a round — about two pieces per sequence on
// average, so that the dispatcher packs rounds of near-equal workgroups instead of waiting for the longest sequence.  Pieces shorter
// than 22 tiles (704 keys) cost more in prologue / merge than they return (pick_splits' measurement).
int decode_plan(const vattn_attn_params* p, const int32_t* lens, vattn_decode_item* items, int cap, int32_t* seq) {
    if (!p || !lens || !items || !seq || p->b <= 0 || p->h_k <= 0 || p->h <= 0) return VATTN_K_ERR_INVALID;
    if (p->seqlen_q != 1 || p->b < 2) return 0;
    const int groups = decode_groups(p);
    if (groups > 1) return 0;                          // MQA groups wider than 32 heads keep the XCD-sibling grid
    const long slots = decode_slots(p);
    const long gps = (long)p->h_k * groups;
    long total = 0, longest = 0;
    for (int b = 0; b < p->b; b++) {
        const long lk = (long)(lens[b] < 0 ? 0 : lens[b]) + p->seqlen_knew;
        const long t = (lk + DC_BN - 1) / DC_BN;
        total += t;
        longest = t > longest ? t : longest;
    }
    if (total <= 0) return 0;
    // what the uniform split would do: every sequence S splits, the launch lasts as long as the longest sequence's share
    vattn_attn_params u = *p;
    u.split_items = nullptr;
    u.num_splits = u.num_splits < 0 ? 0 : u.num_splits;
    const long S = pick_splits(&u, groups, slots);
    const long uniform_makespan = (longest + S - 1) / S;
    // Piece length T: the SMALLEST for which the pieces of all sequences fill whole rounds of the resident workgroups without spilling
    // into another one — R rounds, R = 1 unless the batch alone needs more (then ~1.5 pieces per sequence on average).  [Measured on
    // 256 trace-length sequences, 8 / 1 heads (tools/ragged_decode_probe.py, profiles/r03_ragged_decode_probe.txt): uniform split 52 %
    // of the HBM peak; 647 pieces 64.8 %; 738 pieces — the round of 768 nearly full — 66.4 %; equal lengths 73.5 %.]
    const long rounds = std::max(1L, (3 * (long)p->b * gps + 2 * slots - 1) / (2 * slots));
    const long budget = std::max((long)p->b, rounds * slots / gps);
    auto pieces_at = [&](long t_piece) {
        long n = 0;
        for (int b = 0; b < p->b; b++) {
            const long lk = (long)(lens[b] < 0 ? 0 : lens[b]) + p->seqlen_knew;
            const long t = (lk + DC_BN - 1) / DC_BN;
            n += t > 0 ? (t + t_piece - 1) / t_piece : 1;
        }
        return n;
    };
    long lo = 1, hi = longest > 1 ? longest : 1;        // pieces_at is non-increasing in T; pieces_at(longest) == b <= budget
    while (lo < hi) {
        const long mid = (lo + hi) / 2;
        if (pieces_at(mid) <= budget) hi = mid; else lo = mid + 1;
    }
    long T = lo;
    const long kMinTiles = 22;
    if (T < kMinTiles) T = kMinTiles;
    const bool forced = p->num_splits < 0;              // num_splits = -T: pieces of T tiles whatever the heuristic says (tests, A/B)
    if (forced) T = -(long)p->num_splits;
    if (longest > 128 * T) T = (longest + 127) / 128;   // the merge handles at most 128 pieces per sequence
    // the balanced plan must shorten the launch noticeably: its longest item is T tiles, and it adds partial traffic for every piece
    if (!forced && T * 100 > uniform_makespan * 85) return 0;
    int n = 0;
    for (int b = 0; b < p->b; b++) {
        const long lk = (long)(lens[b] < 0 ? 0 : lens[b]) + p->seqlen_knew;
        const long t = (lk + DC_BN - 1) / DC_BN;
        long cnt = (t + T - 1) / T;
        if (cnt < 1) cnt = 1;
        // equal pieces inside the sequence (not T, T, ..., remainder): the same count, no runt
        const long per = cnt ? (t + cnt - 1) / cnt : 0;
        seq[2 * b] = n;
        seq[2 * b + 1] = (int32_t)cnt;
        for (long j = 0; j < cnt; j++) {
            if (n >= cap) return 0;
            long tb = j * per, te = tb + per;
            if (tb > t) tb = t;
            if (te > t) te = t;
            items[n].b = b;
            items[n].tile_begin = (int32_t)tb;
            // (the last piece of a sequence is open-ended: the kernel clamps to the tiles the DEVICE-side length gives, so stale host
            // lengths cost balance, never keys)
            items[n].tile_end = j == cnt - 1 ? 0x7fffffff : (int32_t)te;
            items[n].index_in_seq = (int32_t)j;
            n++;
        }
    }
    return n;
}

void decode_describe(const vattn_attn_params* p, vattn_plan_desc* out) {
    out->form = 1;
    out->tiling = decode_nb(p);
    const int groups = decode_groups(p);
    if (const int nwg = stream_nwg(p)) {
        out->path = 2;
        out->workgroups = nwg * p->h_k;
        out->merge_launch = 1;
    } else if (p->split_items) {
        out->path = 1;
        out->workgroups = p->num_split_items * p->h_k * groups;
        out->merge_launch = 1;
    } else {
        out->path = 0;
        out->nsplit = pick_splits(p, groups, decode_slots(p));
        out->workgroups = out->nsplit * p->h_k * groups * p->b;
        out->merge_launch = out->nsplit > 1;
    }
}

size_t decode_workspace_bytes(const vattn_attn_params* p) {
    if (const int nwg = stream_nwg(p)) return stream_workspace_bytes(p, nwg);
    if (p->split_items) return (size_t)(p->num_split_items > 0 ? p->num_split_items : 0) * p->h * (p->d + 1) * sizeof(float);
    const int groups = decode_groups(p);
    const int splits = pick_splits(p, groups, decode_slots(p));
    if (splits <= 1) return 0;
    return (size_t)splits * p->b * p->h * (p->d + 1) * sizeof(float);
}

}  // namespace vattn_k
