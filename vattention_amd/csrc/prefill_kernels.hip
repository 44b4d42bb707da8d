// Prefill form (seqlen_q > 1) of flash_attn_with_kvcache on gfx950: chunked causal attention over virtually contiguous K/V.
//   prefill_kernel      : 8 (or 4) waves x 32 query rows per workgroup, 64-key tiles double-buffered in LDS (register-staged
//                         global loads two tiles ahead), S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_{f16,bf16},
//                         softmax entirely in registers (a lane owns one query column), K tile XOR-swizzled for conflict-free
//                         ds_read_b128, V tile as [d-block][key][32 d] sub-tiles read through ds_read_b64_tr_b16;
//                         optional KV split (fp32 partials merged by combine_rows_kernel) and batched variable-length chunks
//   prefill_ilv_kernel  : the same data flow software-pipelined with a hand-written issue order (variant 12)
// Semantics: /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, flash_api.cpp:1291-1578, mask.h:164-196
// (bottom-right causal), softmax.h:69-157 (fp32 max/sum, exp2, P rounded to the I/O dtype before PV).
#include <algorithm>
#include <functional>
#include <queue>
#include <vector>

#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// prefill
// ============================================================================================

}  // namespace vattn_k
#include "prefill_body.h"  // prefill_body / prefill_kernel
namespace vattn_k {


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

// Merge of a work-list launch (vattn_prefill_plan): only the SPLIT query blocks have partials; block sb's share s holds its 256 rows at
// partial rows [part_row + 256 s, +256).  One wave per output row as in combine_rows_kernel; 64 four-wave workgroups per block.
template <typename T, int HD>
__global__ __launch_bounds__(256) void combine_blocks_kernel(vattn_attn_params p) {
    const int lane = threadIdx.x & 63;
    const vattn_prefill_item blk = p.pf_blocks[blockIdx.x >> 6];
    const int r = ((blockIdx.x & 63) << 2) + (threadIdx.x >> 6);        // row inside the 256-row block
    const int b = blk.b, hh = blk.h, q = blk.qb * 256 + r;
    const int sq = p.q_lens ? p.q_lens[b] : p.seqlen_q;
    if (q >= sq) return;
    const int ns = blk.nshares;                                          // <= 16
    const int64_t q_first = p.q_start ? p.q_start[b] : 0;
    const float* oacc = (const float*)p.workspace;
    const float* lacc = oacc + (int64_t)p.pf_part_rows * HD;
    const int64_t row0 = (int64_t)blk.part_row + r;
    const float my = (lane < ns) ? lacc[row0 + 256 * (int64_t)lane] : -INFINITY;
    float mx = my;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, xor_shuffle(mx, o));
    mx = __shfl(mx, 0, 64);
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    const float w = (lane < ns) ? fast_exp2(my - mxs) : 0.f;
    float wsum = w;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) wsum += xor_shuffle(wsum, o);
    wsum = __shfl(wsum, 0, 64);
    const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
    if (2 * lane < HD) {
        const float* src = oacc + row0 * HD + 2 * lane;
        float a0 = 0.f, a1 = 0.f;
        for (int s = 0; s < ns; s++) {
            const float ws = __shfl(w, s, 64);
            const float2 v = *(const float2*)(src + (int64_t)s * 256 * HD);
            a0 += ws * v.x;
            a1 += ws * v.y;
        }
        T* dst = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + q) * p.o_row_stride + (int64_t)hh * p.o_head_stride + 2 * lane;
        dst[0] = Tr<T>::cvt(a0 * inv);
        dst[1] = Tr<T>::cvt(a1 * inv);
    }
    if (p.softmax_lse && lane == 0)
        p.softmax_lse[((int64_t)b * p.h + hh) * p.seqlen_q + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
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
//  tiling  0/1 = 8 waves x 32 rows, 2 = 4 waves x 64 rows (compiler-allocated, spills), 4 = 4 waves x 32 rows, 6 = hand-interleaved
//          8-wave kernel, 7 = prefill64_kernel (prefill64_kernels.hip: 4 waves x 64 rows, LDS-DMA ring, in-wave software pipeline).
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
    if (pl.tiling == 3 || pl.tiling == 5 || (p->d != 128 && (pl.tiling == 2 || pl.tiling == 6 || pl.tiling == 7))) pl.tiling = 1;
    if (pl.tiling == 2 || pl.tiling == 6) pl.tiling = 1;     // lab-only kernels (tools/lab/csrc/prefill_kernels_lab.hip; validate() rejects them before this)
    // keys an average query block sees.  Without a host-side bound the cache VIEW's row count stands in for the lengths, exactly as in
    // FlashAttention's own heuristic (flash_api.cpp:258-323 sizes the split from seqlen_k = k_cache.size(1)); the kernels divide the keys a
    // block REALLY sees (device-side lengths), so an over-estimate costs balance, never correctness.  [Rounds 1-3 assumed seqlen_q here:
    // a chunk on a long prefix then never got its key range split — 393 instead of 900-1050 TFLOP/s on a tensor-parallel shard.]
    const long lk_view = (long)p->seqlen_k + p->seqlen_knew;
    const long lk = p->max_seqlen_k_hint > 0 ? p->max_seqlen_k_hint : (lk_view > p->seqlen_q ? lk_view : p->seqlen_q);
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
    // d = 128: prefill64_kernel (4 waves x 64 rows, one 256-row workgroup per CU, LDS-DMA ring, in-wave software pipeline; +15-20 % on
    // every shape that gives its workgroups enough key tiles to amortise the longer prologue) whenever its grid — after the same
    // KV split as the 8-wave tiling — fills at least 3/4 of the CUs and every workgroup gets >= 24 key tiles.  Measured
    // (tools/kbench.py, profiles/r02_kbench.txt): Yi-6B 32 k prompt 968 -> 1165 TFLOP/s, 16 k chunk @ 112 k 1012 -> 1190,
    // Llama-70B/TP8 2 k chunk @ 30 k 918 -> 1052; short whole prompts (2 k tokens: 16 tiles per workgroup) and grids that stay
    // under 192 workgroups keep the tilings below.
    // Exception (profiles/r02_kbench_split_sweep.txt): a causal WHOLE prompt whose grid is at most one workgroup per CU (Llama-70B/TP8
    // 8 k prompt: 256 workgroups of 4 ... 128 key tiles) is bound by its longest workgroup's key walk; two key-range shares per
    // query block on the 8-wave tiling (two rounds, heaviest first) measure 0.192 ms against 0.239 (prefill64 unsplit), 0.226 (4-wave
    // tiling unsplit) and 0.201 (prefill64, two shares).
    if (p->d == 128 && (uniform || wg8 > 256)) {
        int ns7 = wg8 >= 256 ? 1 : pick(wg8, 256);
        if (uniform && wg8 >= 256) {
            // a few rounds of EQUAL workgroups (a chunk on a long prefix): if the last round is far from full (Yi-34B/TP4: 14 heads x
            // 64 query blocks = 3.5 rounds, 12.5 % of the chip idle at the end), split the key range so that whole rounds come out
            auto eff = [](long w) { return (double)w / (double)(((w + 255) / 256) * 256); };
            double best = eff(wg8);
            for (int ns = 2; ns <= 4 && best < 0.93; ns++)
                if (tiles / ns >= 24 && eff(wg8 * ns) > best + 0.04) { best = eff(wg8 * ns); ns7 = ns; }
        }
        if (wg8 * ns7 >= 192 && tiles / ns7 >= 24) {
            pl.tiling = 7;
            pl.nsplit = ns7;
            return pl;
        }
    }
    if (wg8 > 256) return pl;
    if (wg8 == 256) {
        // ... but only when a share still walks >= 24 key tiles: a 2 k prompt with 32 heads is also exactly 256 workgroups, and there
        // the two-launch split costs more than the imbalance (0.082 ms split vs 0.060 ms on the 4-wave tiling, profiles/r02_kbench.txt)
        const int ns = uniform ? 1 : cap_by_tiles(2);
        if (ns > 1 && tiles / ns >= 24) pl.nsplit = ns;
        else pl.tiling = 4;
        return pl;
    }
    const int ns8 = pick(wg8, 256);
    if (ns8 == 1) { pl.tiling = 4; return pl; }                      // cannot split (short prefix): more, smaller workgroups
    // (a causal whole prompt of at most half a round of 8-wave workgroups — Llama-70B/TP8 4 k: 128 — does better as twice as many
    // 4-wave workgroups with the same shares: 0.074 vs 0.079 ms, tests/test_gpu_plan_gate.py)
    if (wg8 * ns8 >= 192 && (uniform || wg8 > 128)) { pl.nsplit = ns8; return pl; }
    pl.tiling = 4;
    pl.nsplit = pick(wg4, 512);
    return pl;
}

// (the single-launch merge of the key-range shares — variant bits 14 / 15 — measured slower and lives in the lab copy: profiles/r02_kbench_prefill_merge.txt)
template <typename T, int HD, int WAVES, int QC> void launch_prefill(const vattn_attn_params* p, hipStream_t st, int nsplit) {
    constexpr bool MSUM = false;
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
        return true;
    }();
    (void)attr_once;
    hipLaunchKernelGGL((prefill_kernel<T, HD, true, WAVES, QC, MSUM>), grid, block, smem, st, *p, order, nqb, nsplit);
    if (nsplit > 1) {
        const int64_t rows = (int64_t)p->b * p->seqlen_q * p->h;
        hipLaunchKernelGGL((combine_rows_kernel<T, HD>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, *p, nsplit, p->seqlen_q, rows);
    }
}

template <typename T, int HD> int launch_prefill_t(const vattn_attn_params* p, hipStream_t st) {
    if (p->k_new && p->seqlen_knew > 0) launch_append(p, st);
    if constexpr (HD == 128) {
        if (p->pf_items) {        // host-planned work list: prefill64 pieces longest first, then the merge of the split blocks
            if (p->num_pf_items <= 0 || (p->num_pf_blocks > 0 && (!p->pf_blocks || !p->workspace)))
                return fail(VATTN_K_ERR_INVALID, "pf_items needs num_pf_items, and pf_blocks + a workspace when blocks are split");
            // grouped by workgroup (vattn_prefill_plan_wg): persistent workgroups, continuous tile stream (prefill64p_kernels.hip);
            // the fused-RoPE form and outputs without 16-byte rows keep one workgroup per piece (the queue order is a valid list order)
            int* ctr = nullptr;
            bool persistent = p->pf_num_wg > 0 && !p->rotary_cos_sin && ((p->o_row_stride | p->o_head_stride | p->o_batch_stride) & 7) == 0;
            if (persistent && !p->pf_wg_first) {      // drawn queues need the library's counters (none while a graph is being captured before they exist)
                ctr = queue_counters(st);
                // the counters are zeroed IN FRONT OF every drawn launch, stream-ordered (a memset node when captured): a graph replayed
                // on another stream beside eager launches on the capture stream, or a launch that was aborted, must not leave tickets
                // behind for the next one (the kernel's own reset by the last draw stays — it costs nothing)
                persistent = ctr != nullptr && hipMemsetAsync(ctr, 0, 16 * sizeof(int), st) == hipSuccess;
            }
            if (persistent) launch_prefill64p(p, st, ctr);
            else launch_prefill64(p, st, 1);
            if (p->num_pf_blocks > 0) hipLaunchKernelGGL((combine_blocks_kernel<T, 128>), dim3((unsigned)p->num_pf_blocks * 64), dim3(256), 0, st, *p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
            return VATTN_K_OK;
        }
    }
    const PrefillPlan pl = plan_prefill(p);
    if (pl.nsplit > 1 && !p->workspace) return fail(VATTN_K_ERR_INVALID, "KV-split prefill needs a workspace (vattn_attn_workspace_bytes)");
    bool launched = false;
    if constexpr (HD == 128) {
        if (pl.tiling == 7) {
            launch_prefill64(p, st, pl.nsplit);
            if (pl.nsplit > 1) {
                const int64_t rows = (int64_t)p->b * p->seqlen_q * p->h;
                hipLaunchKernelGGL((combine_rows_kernel<T, 128>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, *p, pl.nsplit, p->seqlen_q, rows);
            }
            launched = true;
        }
    }
    if (launched) {
    } else if (pl.tiling == 4) launch_prefill<T, HD, 4, 1>(p, st, pl.nsplit);
    else launch_prefill<T, HD, 8, 1>(p, st, pl.nsplit);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

// Work list of a prefill launch (include/vattn_kernels.h, vattn_prefill_plan).  One 256-row prefill64 workgroup per CU; the launch
// lasts as long as the most loaded CU.  Candidate plans cut every query block longer than T key tiles into ceil(tiles / T) equal pieces,
// for T = longest / {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}; each candidate is priced by replaying the dispatcher (pieces longest first,
// each to the CU that frees up first) with a per-piece cost of tiles + 3 (prologue: Q, the first DMA round trips, the first S')
// + 1.5 for a piece that publishes a partial, plus the merge pass's traffic; the cheapest wins.  [Measured, profiles/r03_kbench.txt: a
// plan whose piece count lands just above a round of 256 — 280 pieces of a 2 k chunk on a 30 k prefix — costs 0.371 ms against 0.279 for
// 256 pieces: pricing whole rounds is what the replay is for.]  Short key walks (no block of 96 tiles = 6 k keys) keep the default
// launch: their time is prologue and merge, and the 4-wave tiling's smaller blocks do better there (2 k prompt, 32 heads: 0.071 vs 0.089).
// wg_first != NULL: the PERSISTENT form (include/vattn_kernels.h, vattn_prefill_plan_wg; csrc/prefill64p_kernels.hip).  A piece that follows
// another one in a workgroup's queue costs its epilogue and a fragment of a tile (2 half-tile units instead of 6), so finer cuts pay off
// (pieces down to 8 tiles) and every list is worth having — also the balanced several-round grids, whose blocks then chain instead of
// being dispatched one by one; the pieces are ASSIGNED here (longest first, each to the least loaded workgroup of its kv head's XCD
// class) and come back grouped by workgroup.
int prefill_worklist(const vattn_attn_params* p, const int32_t* q_lens, const int32_t* k_lens, vattn_prefill_item* items, int cap_items,
                     vattn_prefill_item* blocks, int cap_blocks, int32_t* counts, int32_t* wg_first, int max_wg, int persist_mode) {
    if (!p || !k_lens || !items || !blocks || !counts || p->b <= 0 || p->h <= 0 || p->seqlen_q <= 0) return VATTN_K_ERR_INVALID;
    counts[0] = counts[1] = counts[2] = 0;
    if (persist_mode) counts[3] = 0;
    if (p->d != 128 || p->seqlen_q == 1) return 0;
    const bool persist = persist_mode != 0;
    const long kSlots = persist && max_wg > 0 && max_wg < 256 ? max_wg : 256;      // one prefill64 workgroup per CU
    const long kOvh = persist ? 2 : 6;                 // per-piece overhead in half-tile units (chained / cold prologue)
    const long kMinPiece = persist ? 8 : 12;           // pieces shorter than this are all overhead
    long W = 0, longest = 0, nblk = 0;
    auto tiles_of = [&](int e, int qb) -> long {
        const long sq = q_lens ? q_lens[e] : p->seqlen_q, lk = k_lens[e];
        long n_end = lk;
        if (p->is_causal) { const long lim = (long)qb * 256 + 256 + (lk - sq); n_end = lim < lk ? lim : lk; }
        if (n_end < 0) n_end = 0;
        return (n_end + PF_BN - 1) / PF_BN;
    };
    std::vector<long> blk_tiles;                       // per (entry, query block); every head repeats it
    for (int e = 0; e < p->b; e++) {
        const long sq = q_lens ? q_lens[e] : p->seqlen_q;
        for (int qb = 0; qb < (sq + 255) / 256; qb++) {
            const long t = tiles_of(e, qb);
            blk_tiles.push_back(t);
            W += t * p->h;
            nblk += p->h;
            longest = t > longest ? t : longest;
        }
    }
    const long forced_T = p->num_splits < 0 ? -(long)p->num_splits : 0;      // num_splits = -T: pieces of at most T tiles, no questions asked
    // RAGGED batch of chunks: the default grid is (longest entry's blocks) x heads x entries, and the workgroups of the shorter entries
    // beyond their last block exit at once.  Harmless in number — but the hardware stripes consecutive workgroups of an XCD over its
    // shader engines STATICALLY, so a periodic valid / exit pattern starves part of the chip: two prompts of 23 774 and 5 637 tokens in
    // one launch take 1.60 ms against 1.14 + 0.12 ms launched one by one, while a third (1 000-token) entry — period 3 — brings the
    // launch to 1.18 ms (tools/tp8_prefill_probe.py, profiles/r03_tp8_prefill_probe.txt).  A ragged batch therefore ALWAYS gets a
    // list (valid blocks only, longest first), cut or not.
    bool ragged = false;
    if (q_lens && p->b > 1) {
        long lo = 1L << 40, hi = 0;
        for (int e = 0; e < p->b; e++) {
            const long nq = ((long)q_lens[e] + 255) / 256;
            lo = nq < lo ? nq : lo;
            hi = nq > hi ? nq : hi;
        }
        ragged = lo != hi;
    }
    if (W <= 0 || nblk <= 0 || (!forced_T && !ragged && !persist && longest < 48)) return 0;
    if (persist && nblk > cap_items) return 0;
    const long avg = (W + kSlots - 1) / kSlots;
    // grids of several rounds of workgroups whose longest is no longer than ~a round's share are balanced by the dispatcher's
    // longest-first order already (and, when not ragged, keep the XCD-grouped grid order)
    const bool balanced = nblk >= 4 * kSlots || (nblk >= kSlots && longest * 4 <= avg * 5);
    // (round 5: a "balanced" launch of fewer than four rounds is still priced below — greedy longest-first leaves a one-prompt launch on a
    // tensor-parallel shard 10-20 % above its average load, and cutting only its LONGEST blocks in two brings that back)
    if (!forced_T && !ragged && balanced && !persist && nblk >= 4 * kSlots) return 0;
    // The replay of the dispatcher, priced per candidate.  Every head repeats a block's pieces, so the piece costs are kept as (cost,
    // multiplicity) pairs — a few dozen for a one-prompt launch.  Costs and loads are small integers (half-tile units) and the least
    // loaded CU's load never decreases, so the CUs' loads are a COUNT PER LOAD VALUE with a pointer at the smallest occupied one: k
    // pieces of one cost move k CUs from load m to m + cost in one step.  The loads after every assignment are the same multiset that
    // sorting all pieces and a std::priority_queue of 256 loads produce (equal costs, equal loads are interchangeable), for a
    // twentieth of the host time: the plan is built once per engine iteration in front of layer 0's launch, with the GPU idle (round 5:
    // 0.9-2.7 ms -> 0.05-0.15 ms for one to three prompts on a TP8 rank, tools/plan_time.py).
    std::vector<std::pair<long, long>> cm;             // (cost in half-tile units, how many pieces have it)
    std::vector<int> at_load;                          // CUs per load value
    auto price = [&](long T, long* pieces_out, long* rows_out) -> double {
        cm.clear();
        long rows = 0, pieces = 0, total = 0, maxc = 0;
        for (long t : blk_tiles) {
            long ns = (t + T - 1) / T;
            if (ns < 1) ns = 1;
            if (ns > 16) return 1e30;
            const long per = (t + ns - 1) / ns;
            for (long s_ = 0; s_ < ns; s_++) {
                long tb = s_ * per, te = tb + per;
                if (tb > t) tb = t;
                if (te > t) te = t;
                const long c = 2 * (te - tb) + kOvh + (ns > 1 ? 3 : 0);
                cm.emplace_back(c, (long)p->h);
                total += c * p->h;
                maxc = c > maxc ? c : maxc;
            }
            pieces += ns * p->h;
            if (ns > 1) rows += 256 * ns * p->h;
        }
        std::sort(cm.begin(), cm.end(), std::greater<std::pair<long, long>>());      // longest first
        // (the least loaded CU is never above the final average, total / kSlots: no load exceeds that + the largest cost)
        at_load.assign((size_t)(total / kSlots + maxc + 2), 0);
        at_load[0] = (int)kSlots;
        size_t mn = 0;
        long makespan = 0;
        for (const auto& cmi : cm) {
            long left = cmi.second;
            while (left > 0) {
                while (at_load[mn] == 0) mn++;
                const long k = left < at_load[mn] ? left : at_load[mn];
                at_load[mn] -= (int)k;
                at_load[mn + (size_t)cmi.first] += (int)k;
                const long l = (long)mn + cmi.first;
                makespan = l > makespan ? l : makespan;
                left -= k;
            }
        }
        *pieces_out = pieces;
        *rows_out = rows;
        // merge pass: every partial row is written once and read once (516 B each way) at ~3 TB/s, in half-tile units of ~0.9 us
        return (double)makespan + (double)rows * 1032.0 / 3.0e12 / 0.9e-6;
    };
    // candidate piece lengths T: the longest block cut in {1 .. 16} equal shares, and FRACTIONS of the longest block in between — T = 0.9 x
    // longest cuts only the top tenth of the blocks (in two), which is what evens out the tail of a greedy longest-first schedule at the
    // price of few partials (modelled on the replay's tensor-parallel batches: 4-7 % on one-prompt launches, tools/)
    static const long kShares[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    static const double kFractions[] = {0.95, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.6, 0.55, 0.45, 0.4, 0.36, 0.3};
    std::vector<long> cands;
    for (long ns_max : kShares) {
        const long t_c = (longest + ns_max - 1) / ns_max;
        if (t_c < kMinPiece && ns_max > 1) break;      // pieces shorter than ~12 tiles are all prologue
        cands.push_back(t_c);
    }
    for (double f : kFractions) {
        const long t_c = (long)(longest * f);
        if (t_c >= kMinPiece && t_c * 16 >= longest) cands.push_back(t_c);
    }
    std::sort(cands.begin(), cands.end(), std::greater<long>());
    cands.erase(std::unique(cands.begin(), cands.end()), cands.end());
    double best = 1e30, uncut = 1e30;
    long T = 0, best_rows = 0;
    if (forced_T) {
        long pieces = 0;
        T = forced_T;
        if (longest > 16 * T) T = (longest + 15) / 16;
        if (price(T, &pieces, &best_rows) >= 1e30 || pieces > cap_items) return 0;
    }
    const bool search = !forced_T && nblk < 4 * kSlots && longest >= 48;      // several rounds of blocks / short key walks: nothing to cut
    if (!forced_T && !search) T = longest;
    for (long t_c : cands) {
        if (!search) break;
        long pieces = 0, rows = 0;
        const double c = price(t_c, &pieces, &rows);
        if (t_c == longest) uncut = c;
        if (pieces > cap_items) continue;
        if (c < best - 1e-9) { best = c; T = t_c; best_rows = rows; }
    }
    // a launch the dispatcher balances on its own keeps its blocks whole unless cutting buys at least 2 %
    if (search && balanced && T < longest && best > 0.98 * uncut) { T = longest; best_rows = 0; }
    if (T == 0 || (!forced_T && !ragged && !persist && T >= longest)) return 0;      // nothing worth cutting, nothing to compact: the default launch
    if (best_rows > 0x7fffffffL - 4096) return 0;
    int n = 0, nb = 0;
    long part_rows = 0;
    size_t bi = 0;
    for (int e = 0; e < p->b; e++) {
        const long sq = q_lens ? q_lens[e] : p->seqlen_q;
        for (int qb = 0; qb < (sq + 255) / 256; qb++) {
            const long t = blk_tiles[bi++];
            long ns = (t + T - 1) / T;
            if (ns < 1) ns = 1;
            const long per = (t + ns - 1) / ns;
            for (int j = 0; j < p->h; j++) {
                // the h pieces of one length are neighbours in the list and land on XCD (position % 8): enumerate the heads so that an
                // XCD keeps seeing ONE kv head (the grid orders' rule, attn_common.h wg_to_work), when the head counts allow it
                int h = j;
                if (p->h % 8 == 0 && p->h_k <= 8 && 8 % p->h_k == 0) {
                    const int x = j % 8, r = j / 8, kv = x % p->h_k, G = p->h / p->h_k;
                    h = kv * G + (x / p->h_k) * (p->h / 8) + r;
                }
                if (ns > 1) {
                    if (nb >= cap_blocks) return 0;
                    blocks[nb] = vattn_prefill_item{e, h, qb, 0, 0, (int32_t)ns, (int32_t)part_rows, 0};
                    nb++;
                }
                for (long s_ = 0; s_ < ns; s_++) {
                    if (n >= cap_items) return 0;
                    long tb = s_ * per, te = tb + per;
                    if (tb > t) tb = t;
                    if (te > t) te = t;
                    items[n] = vattn_prefill_item{e, h, qb, (int32_t)tb, (int32_t)te, (int32_t)ns, ns > 1 ? (int32_t)(part_rows + 256 * s_) : -1, s_ == ns - 1 ? 1 : 0};
                    n++;
                }
                if (ns > 1) part_rows += 256 * ns;
            }
        }
    }
    // longest first (stable: pieces of one length keep their (entry, block, head) order, so the heads of a kv group stay neighbours)
    std::stable_sort(items, items + n, [](const vattn_prefill_item& a, const vattn_prefill_item& c) {
        return (a.tile_end - a.tile_begin) > (c.tile_end - c.tile_begin);
    });
    // The list is a PERFORMANCE hint, never a statement about the data: the last share of every query block is open-ended (the kernel
    // clamps every range to the tiles the block really sees, computed from the device-side lengths), so a caller whose host-side lengths
    // are stale gets the right result at a worse balance instead of dropped keys.
    std::vector<long> piece_tiles(n);
    for (int i = 0; i < n; i++) {
        piece_tiles[i] = (long)items[i].tile_end - items[i].tile_begin;
        if (items[i].reserved) items[i].tile_end = 0x7fffffff;
    }
    counts[0] = n;
    counts[1] = nb;
    counts[2] = (int32_t)part_rows;
    if (persist_mode == 2) {
        // drawn queues: the list stays in longest-first order (its head enumeration already deals the kv heads to the XCDs by position);
        // the workgroups take piece w first and draw the rest (csrc/prefill64p_kernels.hip)
        long nwg = n < kSlots ? n : kSlots;
        if (nwg >= 8) nwg -= nwg % 8;
        counts[3] = (int32_t)nwg;
    } else if (persist) {
        // ---- assignment: at most kSlots workgroups; workgroup w runs on XCD w % 8 (a grid of at most one workgroup per CU is handed out
        // round-robin), and an XCD's L2 should keep seeing ONE kv head: the pieces of kv head hk go to the workgroups of class
        // hk % ncls, ncls = the kv heads when they divide the 8 XCDs.  Longest first, each to the least loaded workgroup of its class. ----
        long nwg = n < kSlots ? n : kSlots;
        if (nwg >= 8) nwg -= nwg % 8;
        const int G = p->h / p->h_k;
        const int ncls = (nwg >= 8 && p->h_k <= 8 && 8 % p->h_k == 0) ? p->h_k : 1;
        typedef std::pair<long, int> LW;                                        // (load in half-tile units, workgroup)
        std::vector<std::priority_queue<LW, std::vector<LW>, std::greater<LW>>> heaps(ncls);
        for (int w = 0; w < nwg; w++) heaps[(w % 8) % ncls].push(LW(4, w));     // every queue starts cold: + 4 over the chained overhead
        std::vector<int> owner(n);
        std::vector<int> fill(nwg + 1, 0);
        for (int i = 0; i < n; i++) {
            const long tiles = piece_tiles[i];
            auto& hp = heaps[(items[i].h / G) % ncls];
            LW top = hp.top();
            hp.pop();
            owner[i] = top.second;
            fill[top.second + 1]++;
            top.first += 2 * tiles + kOvh + (items[i].nshares > 1 ? 3 : 0);
            hp.push(top);
        }
        for (int w = 0; w < nwg; w++) fill[w + 1] += fill[w];
        std::vector<vattn_prefill_item> grouped(n);
        std::vector<int> pos(fill.begin(), fill.end() - 1);
        for (int i = 0; i < n; i++) grouped[pos[owner[i]]++] = items[i];        // stable: a queue keeps the longest-first order
        for (int i = 0; i < n; i++) items[i] = grouped[i];
        for (int w = 0; w <= nwg; w++) wg_first[w] = fill[w];
        counts[3] = (int32_t)nwg;
    }
    return n;
}

int launch_prefill_form(const vattn_attn_params* p, hipStream_t st) {
    const bool f16 = p->dtype == VATTN_DTYPE_F16;
    if (p->d == 64) return f16 ? launch_prefill_t<_Float16, 64>(p, st) : launch_prefill_t<__bf16, 64>(p, st);
    return f16 ? launch_prefill_t<_Float16, 128>(p, st) : launch_prefill_t<__bf16, 128>(p, st);
}

void prefill_describe(const vattn_attn_params* p, vattn_plan_desc* out) {
    out->form = 0;
    if (p->pf_items && p->d == 128) {
        out->path = 1;
        out->tiling = 7;
        out->workgroups = p->pf_num_wg > 0 && !p->rotary_cos_sin && ((p->o_row_stride | p->o_head_stride | p->o_batch_stride) & 7) == 0 ? p->pf_num_wg : p->num_pf_items;
        out->merge_launch = p->num_pf_blocks > 0;
        return;
    }
    const PrefillPlan pl = plan_prefill(p);
    out->tiling = pl.tiling == 0 ? 1 : pl.tiling;
    out->nsplit = pl.nsplit;
    const int bm = pl.tiling == 7 ? 256 : pl.tiling == 4 ? 128 : 256;
    out->workgroups = ((p->seqlen_q + bm - 1) / bm) * p->h * p->b * pl.nsplit;
    out->merge_launch = pl.nsplit > 1;
}

size_t prefill_workspace_bytes(const vattn_attn_params* p) {
    if (p->pf_items) return (size_t)(p->pf_part_rows > 0 ? p->pf_part_rows : 0) * (p->d + 1) * sizeof(float);
    const int ns = plan_prefill(p).nsplit;
    return ns > 1 ? (size_t)ns * p->b * p->seqlen_q * p->h * (p->d + 1) * sizeof(float) : 0;
}

}  // namespace vattn_k
