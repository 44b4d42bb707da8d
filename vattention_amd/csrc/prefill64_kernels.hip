// Prefill form of flash_attn_with_kvcache, second-generation kernel for head dimension 128 (gfx950):
//   * a workgroup is 4 waves = one 256-row query block; every wave owns 64 query rows (two 32-row blocks) and a whole SIMD
//     (one wave per SIMD, 512 registers): each K fragment and each V^T fragment read from LDS feeds TWO MFMAs, halving the
//     LDS fragment traffic per flop of the 8 x 32-row kernel (prefill_kernels.hip);
//   * K / V tiles (64 keys) travel HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, one 1-KiB piece per wave
//     instruction), no staging registers: the LDS image of a tile is shaped on the GLOBAL side (lane i fetches the 16 bytes that
//     belong at LDS position M0 + 16*i) — K as 16 padded 4-row pieces whose fragment addresses are lane + immediate (product;
//     the first version's row-major image with the chunk index XOR-swizzled by the row is build 3), V as [d/32][key][32 d]
//     sub-tiles for `ds_read_b64_tr_b16`; a K ring of 2 and a V ring of 3 slots (K runs one tile ahead of V), one barrier per
//     tile, counted `vmcnt` (the DMA is issued by inline asm, the compiler's wait-count pass never sees it);
//   * software pipeline inside the wave: phase A  S(t+1) = K(t+1).Q^T  ||  P(t) = exp2(S(t)), row sums;
//                                        phase B  O += V(t)^T.P(t)^T   ||  row max of S(t+1), f16 packing of P(t);
//     so the softmax VALU work of a tile is issued between the MFMAs of its neighbours by the same wave; the scalar bookkeeping of
//     the DMA stream (running descriptors in fixed SGPR quads, slot rotation) and the row-max reduction sit inside MFMA gaps as
//     well — one instruction before a step's first MFMA, two behind its last (round 3);
//   * softmax exactly as the reference states it (softmax.h:69-94): P = exp2(s*scale*log2e - m*scale*log2e) in fp32, one v_fma
//     + one v_exp per score; the running maximum is only moved when a tile's maximum exceeds it by more than 2^kDeferLog2
//     (deferred rescale, cdna guide T13) — O and l are rescaled exactly once in that (rare) branch, the pending S(t+1) is still
//     raw and needs nothing.  [Measured and dropped: pre-scaling Q (rounds q once more: 2.2x the reference-numerics error on
//     short contexts) and carrying -m in the accumulators (64 extra moves per tile) — with one wave per SIMD the kernel is bound
//     by instruction ISSUE (about 8 slots per MFMA), so the instruction count per tile is what matters.]
// Semantics as prefill_kernels.hip: /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, mask.h:164-196
// (bottom-right causal), softmax.h:69-157 (fp32 max/sum via exp2, P rounded to the I/O dtype before PV),
// flash_fwd_kernel.h:57-499 (the operator's non-split kernel), :1116-1297 (split combine, here combine_rows_kernel).
// Every K/V access is bounded by a buffer descriptor that ends at the sequence's visible length.
#include "prefill64_common.h"

namespace vattn_k {


// The schedule of the tile step (compile-time constants below): 24 of the tile's 32 exp2 pairs start in phase A, a fragment ring of 4
// (three fragments ahead of their MFMA), the row-max chain in phase-B groups 8-23, the per-tile wait + barrier in front of group 8, one
// LDS-DMA piece in groups 9, 12, ... 30.  Every alternative that was measured — other placements, the XOR-swizzled K image, timing
// ablations, per-group clock stamps, the issue-price list of round 6 — lives in the LAB copy of this kernel, tools/lab/csrc/prefill64_lab.hip
// (built into tools/lab/libvattn_lab.so only).
// Round 6 (profiles/r06_p64_price_list.txt, r06_p64_issue_ablation_pmc.txt): the step is bound by what ONE wave can issue to the vector
// ALU — a VALU instruction beside the MFMAs costs 7.3 cycles (v_exp 10.7, packed-f32 43), a scalar instruction, an s_nop or an
// s_waitcnt that does not wait 0.3, and the matrix pipe idles a third of the time.  Hence: (1) nothing per-lane that a scalar
// register can carry stays in the VALU (the scale of the exp2 argument; the distance of an LDS-DMA piece from piece 0 travels in the
// load's scalar offset instead of a v_mad per piece); (2) the first V^T fragments of phase B are read during the LAST groups of phase A
// (the first P.V MFMA used to wait a whole LDS round trip with the matrix pipe idle); (3) the row-max chain starts from its first link.
// What is left per tile is the algorithm's own 64 v_fma + 64 v_exp + 64 v_add + 32 v_cvt_pk + 32 v_max3 and ~20 others against 64 MFMAs.
template <typename T>
__global__ __launch_bounds__(256, 1) void prefill64_kernel(vattn_attn_params p, int order, int nqb, int nsplit) {
    constexpr int NA = 24, RING = 4, MS = 8, BJ = 8, D0 = 9, DS = 3;
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int HD = 128;
    using S = PfSmem<HD>;
    constexpr int BM = 256;
    constexpr int KK = HD / 16;        // k-steps of the S^T MFMA chain
    constexpr int DB = HD / 32;        // 32-wide d blocks of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K ring [2][16 KiB], then V ring [3][16 KiB]; LDS address 0
    // (no static __shared__ in this kernel: the LDS-DMA destinations are ABSOLUTE LDS addresses that assume smem starts at 0; the
    // merge ticket lives in the 16 bytes behind the V ring)
    // The K image in LDS is stored as 16 pieces of 4 rows, each piece 1088 bytes apart (64
    // bytes of padding), inside a piece chunk c of row r3 at byte 64*c + 16*r3.  ds_read_b128's lane groups ({0-3,12-15,20-27}, ...)
    // then hit 16 distinct 16-byte slots of the 256-byte bank row WITHOUT an XOR swizzle, so the address of fragment (kk, kb) is
    // one lane-dependent register + the immediate 8704*kb + 128*kk (+ the slot, static because slots go by (t - tb) & 1 and the
    // loop is unrolled twice): no per-fragment address arithmetic in the hot loop.
    constexpr int KPIECE = 1088;
    constexpr int KSLOT = 16 * 1088;
    constexpr int VBASE = 36864;
    // MS: first phase-B group of the row-max chain of S'(t+1).  BJ: the phase-B group that opens with the per-tile wait + barrier; the
    // eight DMA pieces go out in groups D0, D0 + DS, ... (all >= BJ).
    static_assert(D0 >= BJ && D0 + 7 * DS < 32, "DMA pieces behind the barrier, inside phase B");
    static_assert(NA >= 16 && NA < 32, "key slice 0 of P is packed in phase-A groups 13 / 15: its eight pairs must be exponentiated by group 12");
    static_assert(MS >= 4 && MS + 19 < 32, "row-max chain >= 4 MFMAs behind the last S^T MFMA, its reduction inside phase B");
    auto dma_gap = [](int k) { return D0 + DS * k; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    int b, h, qb, split;
    // host-planned work list (vattn_prefill_plan): blockIdx.x = one piece, longest pieces first; else the grid orders of wg_to_work
    const bool listed = p.pf_items != nullptr;
    int it_tb = 0, it_te = 0, it_row = -1;
    if (listed) {
        const vattn_prefill_item it = p.pf_items[blockIdx.x];
        b = __builtin_amdgcn_readfirstlane(it.b);
        h = __builtin_amdgcn_readfirstlane(it.h);
        qb = __builtin_amdgcn_readfirstlane(it.qb);
        it_tb = __builtin_amdgcn_readfirstlane(it.tile_begin);
        it_te = __builtin_amdgcn_readfirstlane(it.tile_end);
        it_row = __builtin_amdgcn_readfirstlane(it.nshares > 1 ? it.part_row : -1);
        split = 0;
    } else if (!wg_to_work(p, order, nqb, nsplit, b, h, qb, split)) return;
    const bool partial = listed ? it_row >= 0 : nsplit > 1;       // this workgroup publishes an fp32 partial instead of output rows
    const int hk = h / (p.h / p.h_k);                          // GQA: head h uses kv head h / (Hq/Hkv)
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
    Lk = Lk > p.seqlen_k ? p.seqlen_k : Lk;                    // never beyond the cache view's rows
    const int Sq = p.q_lens ? __builtin_amdgcn_readfirstlane(p.q_lens[b]) : p.seqlen_q;
    const int64_t q_first = p.q_start ? (int64_t)__builtin_amdgcn_readfirstlane(p.q_start[b]) : 0;
    const bool causal = p.is_causal != 0;
    const int off = Lk - Sq;                                   // bottom-right alignment (mask.h:164-196)
    const int q_wg0 = qb * BM;
    if (q_wg0 >= Sq) return;                                   // shorter chunk than the grid was sized for (before any barrier)
    const int qw0 = q_wg0 + wave * 64;                         // first query row of this wave

    int n_end = Lk;
    if (causal) n_end = min(Lk, q_wg0 + BM + off);             // last key any row of this block may see, +1
    if (n_end < 0) n_end = 0;
    const int nt_all = (n_end + PF_BN - 1) / PF_BN;
    int tb = 0, nt = nt_all;                                   // this workgroup's key tiles [tb, nt)
    if (listed) {
        tb = min(nt_all, it_tb);
        nt = min(nt_all, it_te);
    } else if (nsplit > 1) {
        const int per = (nt_all + nsplit - 1) / nsplit;
        tb = min(nt_all, split * per);
        nt = min(nt_all, tb + per);
    }
    const T* kbase = uniform_ptr((const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride);
    const T* vbase = uniform_ptr((const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride);
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;

    // ---- DMA addressing (tile-invariant per-lane offsets) ----
    // K piece pc = 4*wave + j holds rows 4*pc .. 4*pc+3: lane i -> row 4*pc + (i & 3), 16-byte chunk i >> 2 of that row
    // V piece pc = 4*wave + j = (d block wave, keys 16*j .. 16*j+15): lane i -> key 16*j + (i >> 2), global chunk 4*wave + (i & 3)
    unsigned koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = 4 * (4 * wave + j) + (lane & 3);
        koff[j] = (unsigned)row * k_rs_bytes + (unsigned)((lane >> 2) << 4);
        const int key = 16 * j + (lane >> 2);
        voff[j] = (unsigned)key * v_rs_bytes + (unsigned)((4 * wave + (lane & 3)) << 4);
    }
    using M = Mfma<T>;
    const unsigned k_lds_wave = (unsigned)(wave * 4 * KPIECE);                 // this wave's four K pieces inside a K slot
    const unsigned v_lds_wave = (unsigned)(VBASE + wave * 4096);               // ... and V pieces inside a V slot
    auto kslot = [&](int t) { return (t - tb) & 1; };                          // K(t)'s slot of the ring
    auto k_rsrc = [&](int t) -> u32x4 {
        int rem = Lk - t * PF_BN;
        rem = rem < 0 ? 0 : (rem > PF_BN ? PF_BN : rem);
        return tile_rsrc(kbase + (int64_t)t * PF_BN * p.k_row_stride, (unsigned)rem * k_rs_bytes);
    };
    auto v_rsrc = [&](int t) -> u32x4 {
        int rem = Lk - t * PF_BN;
        rem = rem < 0 ? 0 : (rem > PF_BN ? PF_BN : rem);
        return tile_rsrc(vbase + (int64_t)t * PF_BN * p.v_row_stride, (unsigned)rem * v_rs_bytes);
    };
    auto dma_k_all = [&](int t) {      // K(t) -> K slot t & 1, this wave's four pieces
        const u32x4 r = k_rsrc(t);
        const unsigned l0 = k_lds_wave + (unsigned)(kslot(t) * KSLOT);
        dma_piece_first(l0, r, koff[0]);
        dma_piece(l0 + KPIECE, r, koff[1]);
        dma_piece(l0 + 2 * KPIECE, r, koff[2]);
        dma_piece(l0 + 3 * KPIECE, r, koff[3]);
    };
    auto dma_v_all = [&](int t) {
        const u32x4 r = v_rsrc(t);
        const unsigned l0 = v_lds_wave + (unsigned)(((t - tb) % 3) * S::kTileBytes);      // prologue only: V(tb) -> slot 0, V(tb+1) -> slot 1
        dma_piece_first(l0, r, voff[0]);
        dma_piece(l0 + 1024, r, voff[1]);
        dma_piece(l0 + 2048, r, voff[2]);
        dma_piece(l0 + 3072, r, voff[3]);
    };

    // ---- prologue ----
    // A key row past the sequence's end must hold FINITE data in the V image (its probability is exactly 0, and 0 x NaN would poison
    // O).  On gfx950 the DMA writes zeros for a lane beyond the descriptor's bound (vattn_selftest_layouts [6]); the kernel does not
    // lean on that: a workgroup whose key range reaches the sequence's ragged last tile zero-fills the V ring first.  Every other
    // workgroup only ever multiplies rows that the DMA fetched (tiles past `nt` are computed into S' and never used) and skips the
    // 48 KiB of LDS writes and the barrier in front of its first fetch (below the noise in time: profiles/r03_p64_prologue_epilogue.txt).
    if (nt * PF_BN > Lk) {
        const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < (3 * S::kTileBytes) / (256 * 16); i++) *(uint4*)(smem + VBASE + (i * 256 + tid) * 16) = z;
        __syncthreads();
    }
    // (asking for V(tb) and K(tb+1) only once Q sits in its registers — so that the wait for Q does not also wait for them — was measured:
    // short pieces lose more on the later K(tb+1) than the first S' gains)
    dma_k_all(tb);
    dma_v_all(tb);
    dma_k_all(tb + 1);

    // Q^T fragments (B operand of S^T = K.Q^T): slot (g, j) <-> d = 16*kk + 8*g + j; pre-scaled into the log2 domain
    const float escale = p.softmax_scale * kLog2e;                      // raw score -> log2 domain
    V8 qf[2][KK];
#pragma unroll
    for (int qc = 0; qc < 2; qc++) {
        const int my_q = qw0 + 32 * qc + l31;
        const T* qptr = (const T*)p.q + (p.q_start ? 0 : (int64_t)b * p.q_batch_stride) + (q_first + my_q) * p.q_row_stride + (int64_t)h * p.q_head_stride;
        V8 raw[KK];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (my_q < Sq) {
                v = *(const uint4*)(qptr + 16 * kk + 8 * g);
            }
            raw[kk] = as_v8<V8>(v);
        }
        if (p.rotary_cos_sin && my_q < Sq) {
            // fused RoPE: query row i sits at position (visible keys - Sq) + i; an element and its partner d + 64 live in the same lane
#pragma unroll
            for (int kk = 0; kk < KK / 2; kk++) {
                V8 c, s;
                rope_load<T>(p, (int64_t)(off + my_q), 16 * kk + 8 * g, c, s);
                rope8<T>(raw[kk], raw[kk + KK / 2], c, s);
            }
        }
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            V8 sc8;
#pragma unroll
            for (int j = 0; j < 8; j++) sc8[j] = raw[kk][j];
            qf[qc][kk] = sc8;
            asm volatile("" : "+a"(qf[qc][kk]));       // materialise the fragment as ONE 4-register accumulator tuple, here
        }
    }

    f32x16 o[DB][2];
#pragma unroll
    for (int i = 0; i < DB; i++)
#pragma unroll
        for (int qc = 0; qc < 2; qc++) o[i][qc] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // -(running max) * softmax_scale * log2e of the lane's query: the addend of the exp2 argument (softmax.h:86-94)
    float nmsub[2];
    // lane-local partial row sums (the other half-lane holds the other 32 keys of every tile), TWO independent accumulators per
    // query block, each touched once per MFMA group at most: with one wave per SIMD a dependent VALU chain stalls the wave, and a
    // stalled wave issues no MFMA either
    float l_acc[2][2];
#pragma unroll
    for (int qc = 0; qc < 2; qc++) {
        nmsub[qc] = 0.f;
#pragma unroll
        for (int a4 = 0; a4 < 2; a4++) l_acc[qc][a4] = 0.f;
    }

    // LDS fragment addressing: one lane-dependent base per tensor + immediate offsets
    const unsigned kfrag_lane = (unsigned)((l31 >> 2) * KPIECE + (l31 & 3) * 16 + g * 64);
    auto kfrag = [&](const char* ksm, int f) -> V8 {                  // f = 2*kk + kb: K rows 32*kb + l31, d = 16*kk + 8*g ..
        const int kk = f >> 1, kb = f & 1;
        return *(const V8*)(ksm + kb * 8 * KPIECE + kk * 128 + kfrag_lane);
    };
    const int i16 = lane & 15, dh = (lane >> 4) & 1;
    const unsigned vfrag_lane = (unsigned)((4 * g + (i16 >> 2)) * 64 + (16 * dh + 4 * (i16 & 3)) * 2);
    auto vfrag = [&](const char* vsm, int f) -> V8 {                  // f = 4*ks + db: keys 16*ks .. 16*ks+15, d block db
        const int ks = f >> 2, db = f & 3;
        const char* a1 = vsm + db * S::kVSubBytes + (16 * ks) * 64 + vfrag_lane;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1 + 8 * 64));
        return join_tr<V8>(lo, hi);
    };
    // masks tile tt's scores in place (ragged end of the sequence / causal diagonal) — wave-uniform decision by the caller
    auto mask_tile = [&](int tt, f32x16 (&s)[2][2]) {
        const int n0 = tt * PF_BN;
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const int my_q = qw0 + 32 * qc + l31;
            const int lim = causal ? min(Lk - 1, my_q + off) : Lk - 1;     // last visible key of this query
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = n0 + 32 * kb + 8 * (r >> 2) + 4 * g + (r & 3);
                    if (key > lim) s[kb][qc][r] = -INFINITY;
                }
        }
    };
    // tile tt needs masking (ragged end of the sequence / causal diagonal of this wave's rows) iff tt >= t_mask:
    // 64 tt + 64 > Lk  <=>  tt >= Lk >> 6;   64 tt + 63 > qw0 + off  <=>  tt >= ((qw0 + off - 63) >> 6) + 1 (arithmetic shift)
    const int t_mask = min(Lk >> 6, causal ? ((qw0 + off - 63) >> 6) + 1 : 0x7fffffff);
    auto needs_mask = [&](int tt) -> bool { return tt >= t_mask; };
    auto row_max = [&](const f32x16 (&s)[2][2], int qc) -> float {
        float m0 = fmaxf(s[0][qc][0], s[1][qc][0]);
#pragma unroll
        for (int r = 1; r < 16; r++) m0 = fmaxf(fmaxf(m0, s[0][qc][r]), s[1][qc][r]);     // v_max3_f32
        return fmaxf(m0, swap_halves(m0));
    };
    // moves the running maximum of query block qc up by delta >= 0 (log2 units, per lane): everything accumulated at the old scale
    // — O and l — is rescaled exactly once (cdna guide T13); scores not yet exponentiated are raw and take the new maximum
    auto raise_max = [&](int qc, float delta) {
        const float alpha = fast_exp2(-delta);
        nmsub[qc] -= delta;
#pragma unroll
        for (int i = 0; i < DB; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][qc][r] *= alpha;
#pragma unroll
        for (int a4 = 0; a4 < 2; a4++) l_acc[qc][a4] *= alpha;
    };
    // P(t) -> the PV B-operand fragment of key slice ks for query block qc: slot (g, j) <-> P registers 8*(ks&1) + j of key block ks>>1
    auto pack_p = [&](const f32x16 (&pt)[2][2], int ks, int qc) -> V8 {
        V8 r;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = X::cvt(pt[ks >> 1][qc][8 * (ks & 1) + j]);
        return r;
    };

    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's pieces of K(tb) landed; V(tb), K(tb+1) may still fly
    __builtin_amdgcn_s_barrier();

    f32x16 sc[2][2];      // S(t): raw scores of the current tile; becomes P(t) in place
    f32x16 sd[2][2];
    {
        const char* ksm = smem + kslot(tb) * KSLOT;
#pragma unroll
        for (int f = 0; f < 2 * KK; f++) {
            const V8 a = kfrag(ksm, f);
#pragma unroll
            for (int qc = 0; qc < 2; qc++) {
                if (f < 2) M::qk_first(sc[f & 1][qc], a, qf[qc][f >> 1]);
                else M::qk_acc(sc[f & 1][qc], a, qf[qc][f >> 1]);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // the last MFMA results are VALU-readable from here
        SCHED_FENCE();
        if (needs_mask(tb)) mask_tile(tb, sc);
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const float mx = row_max(sc, qc);
            nmsub[qc] = (mx == -INFINITY) ? 0.f : -mx * escale;     // softmax.h: a fully masked row keeps a zero reference
        }
    }

    // ---- software pipeline of the softmax VALU work, in units of PAIRS of scores (pair e: key block e>>4, half (e>>3)&1, query
    // block (e>>2)&1, registers 8*half + 2*(e&3), +1 — the order in which the P.V key slices consume them).  Stage E (two
    // v_exp) of pair e sits in group GE(e) of the tile's 64 MFMA groups, stage M (two v_fma: s*scale*log2e - m*scale*log2e) one
    // group earlier, stage A (two v_add into the two row-sum accumulators) one group later: no instruction waits for the one
    // before it (one wave per SIMD: a stalled wave issues no MFMA either).
    auto GE = [](int e) { return e < NA ? 1 + (e * 30) / NA : 33 + ((e - NA) * 17) / (32 - NA); };
#define P64_X0(cur, e) cur[(e) >> 4][((e) >> 2) & 1][8 * (((e) >> 3) & 1) + 2 * ((e) & 3)]
#define P64_X1(cur, e) cur[(e) >> 4][((e) >> 2) & 1][8 * (((e) >> 3) & 1) + 2 * ((e) & 3) + 1]
    // the scale as a REAL scalar register: the compiler satisfies "s"(a float the VALU computed) with a vector register, and the fma then
    // read three vector registers beside a running MFMA (and cost copies)
    const unsigned escale_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, escale));
    auto softmax_stages = [&](int G, f32x16 (&cur)[2][2]) {
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int qc = (e >> 2) & 1;
            if (GE(e) - 1 == G)
                asm("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale_s), "v"(nmsub[qc]));
            if (GE(e) == G) asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)));
            if (GE(e) + 1 == G)      // (v_pk_add_f32: a packed-f32 instruction beside the MFMAs costs 43 cycles, six plain ones)
                asm("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(l_acc[qc][0]), "+v"(l_acc[qc][1]) : "v"(P64_X0(cur, e)), "v"(P64_X1(cur, e)));
        }
    };

    // ---- the DMA stream's scalars, carried across the tile steps and advanced INSIDE MFMA gaps: a lone wave pays an issue slot for
    // every SALU instruction, and whatever sits between the last MFMA of a step and the first of the next is not hidden at all ----
    // rk / rv: descriptors of K(t+2) / V(t+1) at step entry (base, bytes left from the base on); phase B moves them one tile on and
    // fetches K(t+3) / V(t+2) through them
    const unsigned k_tile_b = (unsigned)PF_BN * k_rs_bytes, v_tile_b = (unsigned)PF_BN * v_rs_bytes;
    int k_rows_left = Lk - (tb + 2) * PF_BN, v_rows_left = Lk - (tb + 1) * PF_BN;      // rows of the sequence behind the descriptor's base
    auto bound = [](int rows, unsigned rs) -> unsigned {      // scalar min / max: the compiler's own clamp is a VALU v_med3 (+ a copy back that it cannot do)
        int r;
        asm("s_min_i32 %0, %1, 64\n\ts_max_i32 %0, %0, 0" : "=s"(r) : "s"(rows) : "scc");
        return (unsigned)r * rs;
    };
    const unsigned long long kp0 = (unsigned long long)kbase + (unsigned long long)(tb + 2) * k_tile_b;
    const unsigned long long vp0 = (unsigned long long)vbase + (unsigned long long)(tb + 1) * v_tile_b;
    u32x4 rk = {(unsigned)kp0, (unsigned)(kp0 >> 32) & 0xffffu, bound(k_rows_left, k_rs_bytes), 0x00020000u};
    u32x4 rv = {(unsigned)vp0, (unsigned)(vp0 >> 32) & 0xffffu, bound(v_rows_left, v_rs_bytes), 0x00020000u};
    // byte offsets inside the V ring of V(t)'s slot and of the slot V(t+2) goes to (= the one V(t-1) left): slots go by (t - tb) % 3
    unsigned vs_cur = 0, vs_dma = 2 * S::kTileBytes;
    // One tile step of the wave.  cur holds S(t) on entry and P(t) afterwards, nxt receives S(t+1); kf0 / kf1 hold the first two
    // K(t+1) fragments on entry (read before the previous step ended) and the first two of K(t+2) on exit.
    // Invariants at entry: K(t+1) and V(t) have landed and every wave knows it (the barrier of step t-1); K(t+2) and V(t+1) are
    // in flight.  The barrier of this step opens phase-B group BJ: by then every wave has finished reading K(t+1) (phase A) and V(t-1)
    // (step t-1), so K(t+3) -> slot of K(t+1) and V(t+2) -> slot of V(t-1) may be issued behind it — one piece every DS-th group from
    // group D0 on (back-to-back pieces in the barrier's own group and the seven after it, round 2's placement, measured 1-2 % slower on
    // boxes that are not pinned at their power limit: profiles/r03_p64_schedules.txt).
    auto step = [&](int t, const int par, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], V8& kf0, V8& kf1, V8& kf2) {
        // par = (t - tb) & 1, a literal at both call sites: with the padded K layout every K fragment address folds to lane + immediate
        const int s_cur = par;                                                  // slot of K(t), K(t+2)
        const char* ksm = smem + (s_cur ^ 1) * KSLOT;                           // K(t+1)
        const char* ksm_next = smem + s_cur * KSLOT;                            // K(t+2)
        const char* vsm = smem + VBASE + vs_cur;                                // V(t)
        const unsigned lk0 = k_lds_wave + (unsigned)((s_cur ^ 1) * KSLOT);      // K(t+3) -> the slot K(t+1) leaves
        unsigned lv0 = 0;                                                       // V(t+2)'s pieces of this wave (set in phase B)
        const bool mask_next = needs_mask(t + 1);                               // ragged end / causal diagonal: wave-uniform, the last tiles only
        // Every wave runs the SAME straight-line body for every tile of the workgroup (the barrier makes the waves wait for each other
        // anyway): a tile that lies wholly beyond a wave's causal limit is masked to -inf, contributes P = 0, and leaves the running
        // maximum alone; past the last tile S'(t+1) is computed from a zero-filled K slot and never used.
        // ---------------- 64 groups of { MFMA ; fragment read ahead ; a slice of softmax VALU } ----------------
        // phase A: S'(t+1) = K(t+1).Q^T - m   (32 MFMAs: k-step kk = i>>2, key block (i>>1)&1, query block i&1)
        V8 pf[2][2];         // P(t) fragments of the key slice being multiplied and of the next one
        V8 kf[RING];         // RING - 1 fragments (twice as many MFMAs) ahead of their use
        V8 vf[RING];         // V(t)^T fragments of phase B
        SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int f = i >> 1, qc = i & 1;
            if (f < RING - 1) {
                // the fragments read before the previous step ended sit in accumulator registers
                const V8 a = f == 0 ? kf0 : (f == 1 ? kf1 : kf2);
                if (i < 4) M::qk_first_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
                else M::qk_acc_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
            } else if (i < 4) M::qk_first(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            else M::qk_acc(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            if ((i & 1) == 0 && f + RING - 1 < 2 * KK) kf[(f + RING - 1) % RING] = kfrag(ksm, f + RING - 1);
            softmax_stages(i, cur);
            // key slice 0 of P(t) (pairs 0-7: exponentiated by group GE(7) <= 12 for NA >= 16) is packed HERE, so the first P.V MFMA
            // of phase B does not wait for eight conversions issued right in front of it
            if (i == 13) pf[0][0] = pack_p(cur, 0, 0);
            if (i == 15) pf[0][1] = pack_p(cur, 0, 1);
            // the DMA stream's scalars move one tile on (SALU work, inside gaps)
            if (i == 17) lv0 = v_lds_wave + vs_dma;
            if (i == 19) asm volatile("s_mov_b32 %1, %0\n\ts_add_u32 %0, %0, %2\n\ts_cmp_eq_u32 %0, %3\n\ts_cselect_b32 %0, 0, %0"
                                      : "+s"(vs_cur), "=&s"(vs_dma) : "i"(S::kTileBytes), "i"(3 * S::kTileBytes) : "scc");
            if (i == 21) k_rsrc_advance(rk, k_rows_left, k_tile_b, k_rs_bytes);
            if (i == 23) v_rsrc_advance(rv, v_rows_left, v_tile_b, v_rs_bytes);
            // V(t) landed a step ago: its first fragments are asked for while the last S' MFMAs run (the K ring has stopped reading at i = 24)
            if (i == 26) vf[0] = vfrag(vsm, 0);
            if (i == 28) vf[1] = vfrag(vsm, 1);
            if (i == 30 && RING > 3) vf[2] = vfrag(vsm, 2);
            SCHED_FENCE();
        }
        // phase B: O^T += V(t)^T.P(t)^T   (32 MFMAs: key slice ks = j>>3, d block (j>>1)&3, query block j&1)
        float mx0 = -INFINITY, mx1 = -INFINITY, g0 = -INFINITY, g1 = -INFINITY, grow = -INFINITY;
        SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int f = j >> 1, ks = j >> 3, qc = j & 1;
            if (j == BJ) {
                // this wave's pieces of K(t+2) and V(t+1) (issued one step ago) have landed; behind the barrier everyone's have, and
                // every wave is past its reads of K(t+1) and V(t-1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // (a P fragment is packed at least one MFMA group before its first use: no VALU -> MFMA operand hazard to pad)
            M::pv(o[f & 3][qc], vf[f % RING], pf[ks & 1][qc]);
            if ((j & 1) == 0 && f + RING - 1 < 16) vf[(f + RING - 1) % RING] = vfrag(vsm, f + RING - 1);
            softmax_stages(32 + j, cur);
            // P fragments of key slice ks+1 are packed while slice ks is multiplied (4 cvt_pk per group, groups 4 and 6 of a slice)
            if (ks < 3 && (j & 7) == 4) pf[(ks + 1) & 1][0] = pack_p(cur, ks + 1, 0);
            if (ks < 3 && (j & 7) == 6) pf[(ks + 1) & 1][1] = pack_p(cur, ks + 1, 1);
            // row max of S'(t+1): 2 chains x 16 v_max3, groups MS .. MS+15 (>= 4 MFMAs after the last S^T MFMA was issued); the
            // half-wave exchange and the growth test follow in the next gaps, so that only the branch itself is left behind the
            // step's last MFMA
            if (j >= MS && j < MS + 16) {
                const int r = j - MS;
                if (r == 0) {      // the chain's first link needs no -inf to start from
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mx0) : "v"(nxt[0][0][0]), "v"(nxt[1][0][0]));
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mx1) : "v"(nxt[0][1][0]), "v"(nxt[1][1][0]));
                } else {
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx0) : "v"(nxt[0][0][r]), "v"(nxt[1][0][r]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx1) : "v"(nxt[0][1][r]), "v"(nxt[1][1][r]));
                }
            }
            if (j == MS + 16) mx0 = max_halves(mx0);
            if (j == MS + 17) mx1 = max_halves(mx1);
            if (j == MS + 18) {
                // growth of the row maxima over the running maxima, log2 units (nmsub = -m*scale*log2e; -inf for rows that see nothing here)
                asm("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6" : "=&v"(g0), "=&v"(g1) : "v"(mx0), "v"(mx1), "s"(escale_s), "v"(nmsub[0]), "v"(nmsub[1]));
            }
            if (j == MS + 19) asm("v_max_f32 %0, %1, %2" : "=v"(grow) : "v"(g0), "v"(g1));
            // this wave's four pieces of K(t+3) and of V(t+2): every piece takes piece 0's per-lane offset, its distance from piece 0 (4 K rows |
            // 16 V keys per piece) travels in the load's scalar offset (prefill64_common.h)
            if (j == dma_gap(0)) dma_piece_at<0>(lk0, rk, koff[0]);
            if (j == dma_gap(1)) dma_piece_so<KPIECE, 4>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(2)) dma_piece_so<2 * KPIECE, 8>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(3)) dma_piece_so<3 * KPIECE, 12>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(4)) dma_piece_at<0>(lv0, rv, voff[0]);
            if (j == dma_gap(5)) dma_piece_so<1024, 16>(lv0, rv, voff[0], v_rs_bytes);
            if (j == dma_gap(6)) dma_piece_so<2048, 32>(lv0, rv, voff[0], v_rs_bytes);
            if (j == dma_gap(7)) dma_piece_so<3072, 48>(lv0, rv, voff[0], v_rs_bytes);
            if (j == 27) kf0 = kfrag(ksm_next, 0);          // the next step's first K fragments: K(t+2) is behind the barrier
            if (j == 28) kf1 = kfrag(ksm_next, 1);
            if (j == 29 && RING > 3) kf2 = kfrag(ksm_next, 2);
            SCHED_FENCE();
        }
        if (mask_next) {
            mask_tile(t + 1, nxt);
            mx0 = row_max(nxt, 0);
            mx1 = row_max(nxt, 1);
            g0 = __builtin_fmaf(mx0, escale, nmsub[0]);
            g1 = __builtin_fmaf(mx1, escale, nmsub[1]);
            grow = fmaxf(g0, g1);
        }
        if (__builtin_amdgcn_ballot_w64(grow > kDeferLog2) != 0) {          // rare: a row's maximum grew by > 2^6
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");                        // every PV result has landed in O
            SCHED_FENCE();
            raise_max(0, fmaxf(g0, 0.f));
            raise_max(1, fmaxf(g1, 0.f));
            SCHED_FENCE();
            asm volatile("s_nop 3" ::: "memory");                                    // accvgpr writes -> next MFMA read
        }
    };
    // the loop's entry invariants: K(tb+1), V(tb) landed and known to; K(tb+2), V(tb+1) in flight; first fragments of K(tb+1) read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // also: every wave is done with K(tb) (the prologue's S')
    dma_k_all(tb + 2);
    dma_v_all(tb + 1);
    V8 kfa = kfrag(smem + kslot(tb + 1) * KSLOT, 0), kfb = kfrag(smem + kslot(tb + 1) * KSLOT, 1), kfc = kfrag(smem + kslot(tb + 1) * KSLOT, 2);
    for (int t = tb; t < nt; t += 2) {
        step(t, 0, sc, sd, kfa, kfb, kfc);
        if (t + 1 < nt) step(t + 1, 1, sd, sc, kfa, kfb, kfc);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // trailing DMA retired (nothing may land in LDS of
    SCHED_FENCE();                                                                // a later workgroup); last PV results readable
#undef P64_X0
#undef P64_X1

    // ---- epilogue: O^T[d = 32*db + 8*(r>>2) + 4*g + (r&3)][query] ----
    // (Staging the fp32 partials of a key-range piece through LDS so that every store instruction writes whole 512-byte rows instead of
    // 32 bytes of 32 rows was built and measured in round 3: no gain — the cost of the partials (no-store ablation: 5-19 % of the
    // tensor-parallel launches, profiles/r03_p64_prologue_epilogue.txt) is their volume, not their coalescing.)
    const float sc_ln = p.softmax_scale;
#pragma unroll
    for (int qc = 0; qc < 2; qc++) {
        const int my_q = qw0 + 32 * qc + l31;
        const float l_loc = l_acc[qc][0] + l_acc[qc][1];
        const float l_tot = l_loc + swap_halves(l_loc);
        const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
        const float m_log2 = -nmsub[qc];                      // running max of softmax_scale*log2e*q.k
        // row of the partial buffer that query row q of this block goes to
        auto part_row = [&](int q) -> int64_t {
            return listed ? (int64_t)it_row + (q - q_wg0) : (((int64_t)split * p.b + b) * p.seqlen_q + q) * p.h + h;
        };
        float* lpart = (float*)p.workspace + (listed ? (int64_t)p.pf_part_rows : (int64_t)nsplit * p.b * p.seqlen_q * p.h) * HD;
        if (my_q < Sq && partial) {
            const int64_t row = part_row(my_q);
            float* opart = (float*)p.workspace + row * HD;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int tq = 0; tq < 4; tq++) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = o[db][qc][4 * tq + e] * inv;
                    *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                }
            if (g == 0) {
                const float lv = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_log2 + __log2f(l_tot));
                lpart[row] = lv;
            }
        } else if (my_q < Sq) {
            T* optr = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + my_q) * p.o_row_stride + (int64_t)h * p.o_head_stride;
            if (((p.o_row_stride | p.o_head_stride | p.o_batch_stride) & 7) == 0) {
                // 16-byte stores: half-lane pairs exchange 8-byte groups through v_permlane32_swap (see prefill_kernels.hip)
#pragma unroll
                for (int db = 0; db < DB; db++)
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {
                        typename X::v4 we, wo;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            we[e] = X::cvt(o[db][qc][4 * (2 * pr) + e] * inv);
                            wo[e] = X::cvt(o[db][qc][4 * (2 * pr + 1) + e] * inv);
                        }
                        uint2 ue, uo;
                        __builtin_memcpy(&ue, &we, 8);
                        __builtin_memcpy(&uo, &wo, 8);
                        const auto r0 = __builtin_amdgcn_permlane32_swap(ue.x, uo.x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(ue.y, uo.y, false, false);
                        *(uint4*)(optr + 32 * db + 8 * (2 * pr + g)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
            } else {
#pragma unroll
                for (int db = 0; db < DB; db++)
#pragma unroll
                    for (int tq = 0; tq < 4; tq++) {
                        typename X::v4 w;
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = X::cvt(o[db][qc][4 * tq + e] * inv);
                        *(typename X::v4*)(optr + 32 * db + 8 * tq + 4 * g) = w;
                    }
            }
            if (p.softmax_lse && g == 0) {
                // natural-log LSE of scale*QK^T; +inf for fully masked rows (flash convention)
                const float lse = (l_tot == 0.f) ? INFINITY : (m_log2 + __log2f(l_tot)) * 0.6931471805599453f;
                p.softmax_lse[((int64_t)b * p.h + h) * p.seqlen_q + my_q] = lse;
            }
        }
    }
    (void)sc_ln;
}

// host side: grid as prefill_kernels.hip's 1-D / 3-D orders with 256-row query blocks
dim3 prefill_grid(const vattn_attn_params* p, int nqb, int* order_out);       // prefill_kernels.hip

constexpr int kSmem64 = 36864 + 3 * PfSmem<128>::kTileBytes;      // K ring (2 x 17 408, rounded up) + V ring
template <typename T> static void launch64_t(const vattn_attn_params* p, hipStream_t st, int nsplit) {
    const int nqb = (p->seqlen_q + 255) / 256;
    int order;
    dim3 grid = prefill_grid(p, nqb, &order);
    if (p->pf_items) grid = dim3((unsigned)p->num_pf_items);      // one workgroup per listed piece
    else if (nsplit > 1) {
        if (order == 0) {
            vattn_attn_params q = *p;
            q.variant = (p->variant & ~(3 << 5)) | (2 << 5);
            grid = prefill_grid(&q, nqb, &order);
        }
        grid = dim3(((grid.x + 7) / 8) * 8 * nsplit);
    }
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)prefill64_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem64 + 16);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((prefill64_kernel<T>), grid, dim3(256), kSmem64 + 16, st, *p, order, nqb, nsplit);
}

// ONE build per dtype; key-range shares are merged by combine_rows_kernel / combine_blocks_kernel in a second launch (prefill_kernels.hip).
void launch_prefill64(const vattn_attn_params* p, hipStream_t st, int nsplit) {
    if (p->dtype == VATTN_DTYPE_BF16) launch64_t<__bf16>(p, st, nsplit);
    else launch64_t<_Float16>(p, st, nsplit);
}

}  // namespace vattn_k
