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

// LAB (round 6, closed): l(agpr, 4 equal registers) += the lane's OWN four P values — v_mfma_f32_4x4x4 (16 blocks of 4 lanes, K = 4) with A = ones
// gives D[i][j] = sum_k B[k][j], and lane (block, j) holds exactly B[0..3][j]: row sums on the matrix pipe
template <typename T> __device__ __forceinline__ void rowsum4(f32x4& l, typename Tr<T>::v4 ones, typename Tr<T>::v4 b);
template <> __device__ __forceinline__ void rowsum4<_Float16>(f32x4& l, Tr<_Float16>::v4 ones, Tr<_Float16>::v4 b) {
    asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(b));
}
template <> __device__ __forceinline__ void rowsum4<__bf16>(f32x4& l, Tr<__bf16>::v4 ones, Tr<__bf16>::v4 b) {
    asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(b));
}


// ABL bits 0-5: timing ablations for tools/kbench.py (results are WRONG when any is set): bit 0 no LDS-DMA in the steady state,
// bit 1 no exp2 / row sums, bit 2 no row max, bit 3 no per-tile wait + barrier, bit 4 fragment reads only for the first MFMAs of
// a phase, bit 5 no f16 packing.  Bit 7 selects the padded K image (the product instantiates ABL = 128; 0 = round 2's XOR-swizzled
// image, lab).  [Round 2's bit 6 — row sums by v_dot2c over the packed P — measured +-0 (the dot instructions serialise with the
// MFMA pipe, profiles/r02_issue_probe.txt, r02_prefill64_ablations.md) and was removed in round 3.]
// NA: exp2 pairs (of the tile's 32) whose stages start in phase A, the rest in phase B.  RING: K / V^T fragment registers in flight
// (RING - 1 fragments ahead of their MFMA).
// R6 (round 6): bit 0 = the first three V^T fragments of phase B are read inside the LAST groups of phase A (into the K fragment ring's freed
// registers) instead of in front of the first P.V MFMA, which then waited a whole LDS round trip with the matrix pipe idle; bit 1 (LAB) =
// s_memtime stamps at every 8th group of the first 64 tile steps of every wave (tools/lab/p64_stamps.py; behind softmax_lse).
constexpr int kStampBase = 86528;      // LAB: LDS offset of the stamps (behind the K ring, the V ring and the merge ticket), 8 KiB
// XTRA (LAB price list, results unchanged): ONE extra instruction of a class behind every MFMA of the tile step (64 per tile): 1 s_nop 0,
// 2 s_mov (SALU), 3 s_waitcnt that waits for nothing, 4 v_mov (VALU, 4 bytes), 5 v_exp, 6 v_max3 (VALU, 8 bytes, 3 operands), 7 v_pk_fma_f32,
// 8 v_pk_add_f32, 9 v_add_f32, 10 v_cvt_pk_f16_f32, 11 v_pk_mul_f32
template <typename T, int ABL, int NA, int RING, int MS = 8, int BJ = 8, int D0 = 9, int DS = 3, int R6 = 0, int XTRA = 0>
__global__ __launch_bounds__(256, 1) void prefill64_kernel(vattn_attn_params p, int order, int nqb, int nsplit, int* done, int merge_mode) {
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
    // ABL bit 7 (a layout, not an ablation): the K image in LDS is stored as 16 pieces of 4 rows, each piece 1088 bytes apart (64
    // bytes of padding), inside a piece chunk c of row r3 at byte 64*c + 16*r3.  ds_read_b128's lane groups ({0-3,12-15,20-27}, ...)
    // then hit 16 distinct 16-byte slots of the 256-byte bank row WITHOUT an XOR swizzle, so the address of fragment (kk, kb) is
    // one lane-dependent register + the immediate 8704*kb + 128*kk (+ the slot, static because slots go by (t - tb) & 1 and the
    // loop is unrolled twice): no per-fragment address arithmetic in the hot loop.
    constexpr bool KP = (ABL & 128) != 0;
    constexpr int KPIECE = KP ? 1088 : 1024;
    constexpr int KSLOT = KP ? 16 * 1088 : S::kTileBytes;
    constexpr int VBASE = KP ? 36864 : 2 * S::kTileBytes;
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
    // K piece pc = 4*wave + j holds rows 4*pc .. 4*pc+3: lane i -> row 4*pc + (i >> 4), LDS chunk i & 15 <- global chunk (i & 15) ^ (row & 15)
    // V piece pc = 4*wave + j = (d block wave, keys 16*j .. 16*j+15): lane i -> key 16*j + (i >> 2), global chunk 4*wave + (i & 3)
    unsigned koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = 4 * (4 * wave + j) + (KP ? (lane & 3) : (lane >> 4));
        koff[j] = (unsigned)row * k_rs_bytes + (unsigned)((KP ? (lane >> 2) : ((lane & 15) ^ (row & 15))) << 4);
        const int key = 16 * j + (lane >> 2);
        voff[j] = (unsigned)key * v_rs_bytes + (unsigned)((4 * wave + (lane & 3)) << 4);
    }
    using M = Mfma<T>;
    const unsigned k_lds_wave = (unsigned)(wave * 4 * KPIECE);                 // this wave's four K pieces inside a K slot
    const unsigned v_lds_wave = (unsigned)(VBASE + wave * 4096);               // ... and V pieces inside a V slot
    auto kslot = [&](int t) { return KP ? ((t - tb) & 1) : (t & 1); };         // K(t)'s slot of the ring
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

    if constexpr ((R6 & 2) != 0) {
        for (int i = tid; i < 2048; i += 256) ((unsigned*)(smem + kStampBase))[i] = 0u;
        __syncthreads();
    }
    // ---- prologue ----
    // A key row past the sequence's end must hold FINITE data in the V image (its probability is exactly 0, and 0 x NaN would poison
    // O).  On gfx950 the DMA writes zeros for a lane beyond the descriptor's bound (vattn_selftest_layouts [6]); the kernel does not
    // lean on that: a workgroup whose key range reaches the sequence's ragged last tile zero-fills the V ring first.  Every other
    // workgroup only ever multiplies rows that the DMA fetched (tiles past `nt` are computed into S' and never used) and skips the
    // 48 KiB of LDS writes and the barrier in front of its first fetch (below the noise in time: profiles/r03_p64_prologue_epilogue.txt).
    if (!(ABL & 512) && nt * PF_BN > Lk) {
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
                if constexpr ((ABL & 1024) != 0) {      // LAB A/B: Q read once per workgroup — streaming (nt), so that it does not evict the K/V the XCD shares
                    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
                    const nt4 t = __builtin_nontemporal_load((const nt4*)(qptr + 16 * kk + 8 * g));
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                } else {
                    v = *(const uint4*)(qptr + 16 * kk + 8 * g);
                }
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

    // R6 bit 8: the row sums on the MATRIX pipe.  The tile step is bound by what the wave can ISSUE to the vector ALU (profiles/
    // r06_p64_price_list.txt: every VALU instruction beside the MFMAs costs 7.3 cycles, a scalar instruction 0.3, and the matrix pipe idles
    // a third of the time).  v_mfma_f32_4x4x4 (16 independent 4 x 4 blocks of four lanes, K = 4) with A = ones adds the four values a lane
    // holds in its B operand into that lane's accumulator: 16 of them per tile (one per four packed probabilities) replace the tile's 64
    // v_add_f32, lane-local like the adds were (the other half-wave holds the other 32 keys; joined in the epilogue).  The sum is over the
    // probabilities ROUNDED to the I/O dtype — the values the P.V product uses; softmax.h:135-157 sums them before rounding: the difference
    // is the mean of the roundings, ~2^-12 / sqrt(keys) relative for f16.
    constexpr bool MSUM = (R6 & 256) != 0;
    using V4 = typename X::v4;
    f32x4 lsum[2];
    V4 ones4;
    if constexpr (MSUM) {
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            lsum[qc] = (f32x4){0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(lsum[qc]));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) ones4[j] = X::cvt(1.0f);
        asm volatile("" : "+v"(ones4));
    }
    // LDS fragment addressing: one lane-dependent base per tensor + immediate offsets
    const unsigned kfrag_lane = KP ? (unsigned)((l31 >> 2) * KPIECE + (l31 & 3) * 16 + g * 64) : (unsigned)(l31 * S::kRowBytes);
    const unsigned kswz = (unsigned)(l31 & 15);
    auto kfrag = [&](const char* ksm, int f) -> V8 {                  // f = 2*kk + kb: K rows 32*kb + l31, d = 16*kk + 8*g ..
        const int kk = f >> 1, kb = f & 1;
        if (KP) return *(const V8*)(ksm + kb * 8 * KPIECE + kk * 128 + kfrag_lane);
        return *(const V8*)(ksm + kb * 32 * S::kRowBytes + kfrag_lane + (((unsigned)(2 * kk + g) ^ kswz) << 4));
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
        if constexpr (MSUM) {
#pragma unroll
            for (int r = 0; r < 4; r++) lsum[qc][r] *= alpha;
        }
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
    // R6 bit 2: the scale is a REAL scalar register (the compiler satisfies "s"(float computed by the VALU) with a vector register: the
    // fma then reads three vector registers beside a running MFMA); bit 3: stage order M, E, A inside a group instead of A, E, M
    const unsigned escale_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, escale));
    // R6 bits 4-5 (LAB, WRONG results: the maximum is not subtracted): the v_fma replaced by a v_mul in the 4-byte VOP2 encoding / in the
    // 8-byte VOP3 encoding — does the ENCODING SIZE price an instruction beside the MFMAs?  Bit 6 (exact): the same fused multiply-add
    // from 4-byte instructions only: v_mov tmp, -m ; v_fmac tmp, scale, s (VOP2) ; then v_exp s, tmp.
    float tmpE[32][2];
    auto stage_m = [&](f32x16 (&cur)[2][2], int e, int qc) {
        if (ABL & 2048) return;
        if constexpr ((R6 & 16) != 0)
            asm("v_mul_f32_e32 %0, %2, %0\n\tv_mul_f32_e32 %1, %2, %1" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale_s));
        else if constexpr ((R6 & 32) != 0)
            asm("v_mul_f32_e64 %0, %2, %0\n\tv_mul_f32_e64 %1, %2, %1" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale_s));
        else if constexpr ((R6 & 64) != 0)
            asm("v_mov_b32_e32 %0, %4\n\tv_mov_b32_e32 %1, %4\n\tv_fmac_f32_e32 %0, %5, %2\n\tv_fmac_f32_e32 %1, %5, %3"
                : "=&v"(tmpE[e][0]), "=&v"(tmpE[e][1]) : "v"(P64_X0(cur, e)), "v"(P64_X1(cur, e)), "v"(nmsub[qc]), "s"(escale_s));
        else if constexpr ((R6 & 4) != 0)
            asm("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale_s), "v"(nmsub[qc]));
        else
            asm("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale), "v"(nmsub[qc]));
    };
    auto softmax_stages = [&](int G, f32x16 (&cur)[2][2]) {
        if (ABL & 2) return;
        if constexpr ((R6 & 8) != 0) {
#pragma unroll
            for (int e = 0; e < 32; e++)
                if (GE(e) - 1 == G) stage_m(cur, e, (e >> 2) & 1);
        }
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int qc = (e >> 2) & 1;
            if constexpr ((R6 & 8) == 0) {
                if (GE(e) - 1 == G) stage_m(cur, e, qc);
            }
            if (GE(e) == G) {
                if constexpr ((R6 & 64) != 0) asm("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(P64_X0(cur, e)), "=v"(P64_X1(cur, e)) : "v"(tmpE[e][0]), "v"(tmpE[e][1]));
                else asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)));
            }
            if (GE(e) + 1 == G && !(ABL & 4096) && !MSUM)      // (v_pk_add_f32 was tried: forming the register pairs costs more moves than the adds it saves)
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
    // LAB (R6 bit 1): per wave 64 steps x 8 stamps of s_memtime (low 32 bits), kept in LDS behind the rings (a global store per stamp
    // would sit in front of the step's vmcnt(0)) and copied out behind softmax_lse's rows when the workgroup ends
    int ts_step = 0;
    unsigned* const ts_lds = (unsigned*)(smem + kStampBase) + wave * 512;
    auto stamp = [&](int k) {
        if constexpr ((R6 & 2) != 0) {
            if (ts_step < 64) {
                const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) ts_lds[ts_step * 8 + k] = tnow;
            }
        }
    };
    float xdummy = 1.0f;
    f32x2 xdummy2 = {1.0f, 1.0f};
    auto extra = [&]() {
        if constexpr (XTRA == 1) asm volatile("s_nop 0");
        else if constexpr (XTRA == 2) { unsigned d; asm volatile("s_mov_b32 %0, 0" : "=s"(d)); }
        else if constexpr (XTRA == 3) asm volatile("s_waitcnt lgkmcnt(15)");
        else if constexpr (XTRA == 4) asm volatile("v_mov_b32_e32 %0, 1.0" : "=v"(xdummy));
        else if constexpr (XTRA == 5) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(xdummy));
        else if constexpr (XTRA == 6) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(xdummy));
        else if constexpr (XTRA == 7) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(xdummy2));
        else if constexpr (XTRA == 8) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(xdummy2));
        else if constexpr (XTRA == 9) asm volatile("v_add_f32_e32 %0, %0, %0" : "+v"(xdummy));
        else if constexpr (XTRA == 10) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(xdummy));
        else if constexpr (XTRA == 11) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(xdummy2));
    };
    auto step = [&](int t, const int par, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], V8& kf0, V8& kf1, V8& kf2) {
        // par = (t - tb) & 1, a literal at both call sites: with the padded K layout every K fragment address folds to lane + immediate
        const int s_cur = KP ? par : (t & 1);                                   // slot of K(t), K(t+2)
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
        constexpr bool VPRE = (R6 & 1) != 0;
        SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int f = i >> 1, qc = i & 1;
            if ((i & 7) == 0) stamp(i >> 3);
            if (f < RING - 1) {
                // the fragments read before the previous step ended sit in accumulator registers
                const V8 a = f == 0 ? kf0 : (f == 1 ? kf1 : kf2);
                if (i < 4) M::qk_first_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
                else M::qk_acc_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
            } else if (i < 4) M::qk_first(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            else M::qk_acc(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            extra();
            if ((i & 1) == 0 && f + RING - 1 < 2 * KK && !((ABL & 16) && f >= 1)) kf[(f + RING - 1) % RING] = kfrag(ksm, f + RING - 1);
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
            if (VPRE && i == 26) vf[0] = vfrag(vsm, 0);
            if (VPRE && i == 28) vf[1] = vfrag(vsm, 1);
            if (VPRE && i == 30 && RING > 3) vf[2] = vfrag(vsm, 2);
            SCHED_FENCE();
        }
        // phase B: O^T += V(t)^T.P(t)^T   (32 MFMAs: key slice ks = j>>3, d block (j>>1)&3, query block j&1)
        if (!VPRE) {
            vf[0] = vfrag(vsm, 0);
            vf[1] = vfrag(vsm, 1);
            if (RING > 3) vf[2] = vfrag(vsm, 2);
        }
        float mx0 = -INFINITY, mx1 = -INFINITY, g0 = -INFINITY, g1 = -INFINITY, grow = -INFINITY;
        SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int f = j >> 1, ks = j >> 3, qc = j & 1;
            if ((j & 7) == 0) stamp(4 + (j >> 3));
            if (j == BJ && !(ABL & 8)) {
                // this wave's pieces of K(t+2) and V(t+1) (issued one step ago) have landed; behind the barrier everyone's have, and
                // every wave is past its reads of K(t+1) and V(t-1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // (a P fragment is packed at least one MFMA group before its first use: no VALU -> MFMA operand hazard to pad)
            M::pv(o[f & 3][qc], vf[f % RING], pf[ks & 1][qc]);
            extra();
            if ((j & 1) == 0 && f + RING - 1 < 16 && !((ABL & 16) && f >= 1)) vf[(f + RING - 1) % RING] = vfrag(vsm, f + RING - 1);
            softmax_stages(32 + j, cur);
            // row sums of key slice ks (its P fragments were packed at least two groups ago): four 4x4x4 MFMAs per slice, behind the
            // first fillers of a group so that they do not queue straight behind the group's own MFMA
            if constexpr (MSUM) {      // groups 0, 2 of a slice: query block 0 (low, high half of its fragment); 4, 6: block 1
                if ((j & 7) == 0) rowsum4<T>(lsum[0], ones4, __builtin_shufflevector(pf[ks & 1][0], pf[ks & 1][0], 0, 1, 2, 3));
                if ((j & 7) == 2) rowsum4<T>(lsum[0], ones4, __builtin_shufflevector(pf[ks & 1][0], pf[ks & 1][0], 4, 5, 6, 7));
                if ((j & 7) == 4) rowsum4<T>(lsum[1], ones4, __builtin_shufflevector(pf[ks & 1][1], pf[ks & 1][1], 0, 1, 2, 3));
                if ((j & 7) == 6) rowsum4<T>(lsum[1], ones4, __builtin_shufflevector(pf[ks & 1][1], pf[ks & 1][1], 4, 5, 6, 7));
            }
            // P fragments of key slice ks+1 are packed while slice ks is multiplied (4 cvt_pk per group, groups 4 and 6 of a slice)
            if (!(ABL & 32)) {
                if (ks < 3 && (j & 7) == 4) pf[(ks + 1) & 1][0] = pack_p(cur, ks + 1, 0);
                if (ks < 3 && (j & 7) == 6) pf[(ks + 1) & 1][1] = pack_p(cur, ks + 1, 1);
            }
            // row max of S'(t+1): 2 chains x 16 v_max3, groups MS .. MS+15 (>= 4 MFMAs after the last S^T MFMA was issued); the
            // half-wave exchange and the growth test follow in the next gaps, so that only the branch itself is left behind the
            // step's last MFMA
            if (!(ABL & 4) && j >= MS && j < MS + 16) {
                const int r = j - MS;
                if (r == 0 && (R6 & 512) != 0) {      // the chain's first link needs no -inf to start from (two v_mov fewer per tile)
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mx0) : "v"(nxt[0][0][0]), "v"(nxt[1][0][0]));
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mx1) : "v"(nxt[0][1][0]), "v"(nxt[1][1][0]));
                } else if constexpr ((R6 & 128) != 0) {      // two 4-byte v_max instead of one 8-byte v_max3 (exact either way)
                    asm("v_max_f32_e32 %0, %0, %1\n\tv_max_f32_e32 %0, %0, %2" : "+v"(mx0) : "v"(nxt[0][0][r]), "v"(nxt[1][0][r]));
                    asm("v_max_f32_e32 %0, %0, %1\n\tv_max_f32_e32 %0, %0, %2" : "+v"(mx1) : "v"(nxt[0][1][r]), "v"(nxt[1][1][r]));
                } else {
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx0) : "v"(nxt[0][0][r]), "v"(nxt[1][0][r]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx1) : "v"(nxt[0][1][r]), "v"(nxt[1][1][r]));
                }
            }
            if (j == MS + 16) mx0 = max_halves(mx0);
            if (j == MS + 17) mx1 = max_halves(mx1);
            if (j == MS + 18) {
                // growth of the row maxima over the running maxima, log2 units (nmsub = -m*scale*log2e; -inf for rows that see nothing here)
                asm("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6" : "=&v"(g0), "=&v"(g1) : "v"(mx0), "v"(mx1), "s"(escale), "v"(nmsub[0]), "v"(nmsub[1]));
            }
            if (j == MS + 19) asm("v_max_f32 %0, %1, %2" : "=v"(grow) : "v"(g0), "v"(g1));
            if (!(ABL & 1)) {
                if (j == dma_gap(0)) dma_piece_at<0>(lk0, rk, koff[0]);
                // piece j's per-lane offset = piece 0's + j x (4 K rows | 16 V keys): one v_mad beside the DMA instead of six more
                // loop-invariant registers that the allocator parks in the accumulator file and reads back every tile
                if constexpr ((R6 & 512) != 0 && KP) {      // the pieces' distances in the scalar offset (prefill64_common.h)
                    if (j == dma_gap(1)) dma_piece_so<KPIECE, 4>(lk0, rk, koff[0], k_rs_bytes);
                    if (j == dma_gap(2)) dma_piece_so<2 * KPIECE, 8>(lk0, rk, koff[0], k_rs_bytes);
                    if (j == dma_gap(3)) dma_piece_so<3 * KPIECE, 12>(lk0, rk, koff[0], k_rs_bytes);
                    if (j == dma_gap(4)) dma_piece_at<0>(lv0, rv, voff[0]);
                    if (j == dma_gap(5)) dma_piece_so<1024, 16>(lv0, rv, voff[0], v_rs_bytes);
                    if (j == dma_gap(6)) dma_piece_so<2048, 32>(lv0, rv, voff[0], v_rs_bytes);
                    if (j == dma_gap(7)) dma_piece_so<3072, 48>(lv0, rv, voff[0], v_rs_bytes);
                } else {
                if (j == dma_gap(1)) dma_piece_at<KPIECE>(lk0, rk, KP ? piece_off<4>(koff[0], k_rs_bytes) : koff[1]);
                if (j == dma_gap(2)) dma_piece_at<2 * KPIECE>(lk0, rk, KP ? piece_off<8>(koff[0], k_rs_bytes) : koff[2]);
                if (j == dma_gap(3)) dma_piece_at<3 * KPIECE>(lk0, rk, KP ? piece_off<12>(koff[0], k_rs_bytes) : koff[3]);
                if (j == dma_gap(4)) dma_piece_at<0>(lv0, rv, voff[0]);
                if (j == dma_gap(5)) dma_piece_at<1024>(lv0, rv, piece_off<16>(voff[0], v_rs_bytes));
                if (j == dma_gap(6)) dma_piece_at<2048>(lv0, rv, piece_off<32>(voff[0], v_rs_bytes));
                if (j == dma_gap(7)) dma_piece_at<3072>(lv0, rv, piece_off<48>(voff[0], v_rs_bytes));
                }
            }
            if (j == 27) kf0 = kfrag(ksm_next, 0);          // the next step's first K fragments: K(t+2) is behind the barrier
            if (j == 28) kf1 = kfrag(ksm_next, 1);
            if (j == 29 && RING > 3) kf2 = kfrag(ksm_next, 2);
            SCHED_FENCE();
        }
        if constexpr ((R6 & 2) != 0) ts_step++;
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
    if constexpr ((R6 & 2) != 0) {
        __syncthreads();
        unsigned* const ts_out = (unsigned*)(p.softmax_lse + (((size_t)p.b * p.h * p.seqlen_q + 1) & ~(size_t)1)) + (size_t)blockIdx.x * 2048;
        for (int i = tid; i < 2048; i += 256) ts_out[i] = ((unsigned*)(smem + kStampBase))[i];
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
        const float l_own = MSUM ? lsum[qc][0] : l_loc;
        const float l_tot = l_own + swap_halves(l_own);
        const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
        const float m_log2 = -nmsub[qc];                      // running max of softmax_scale*log2e*q.k
        // row of the partial buffer that query row q of this block goes to
        auto part_row = [&](int q) -> int64_t {
            return listed ? (int64_t)it_row + (q - q_wg0) : (((int64_t)split * p.b + b) * p.seqlen_q + q) * p.h + h;
        };
        float* lpart = (float*)p.workspace + (listed ? (int64_t)p.pf_part_rows : (int64_t)nsplit * p.b * p.seqlen_q * p.h) * HD;
        if (ABL & 256) {
            if (l_tot == 12345.f) ((float*)p.workspace)[lane] = o[0][qc][0] * inv;      // ablation: no epilogue stores
        } else if (my_q < Sq && partial) {
            const int64_t row = part_row(my_q);
            float* opart = (float*)p.workspace + row * HD;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int tq = 0; tq < 4; tq++) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = o[db][qc][4 * tq + e] * inv;
                    if (kLab && merge_mode == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) store_dev(opart + 32 * db + 8 * tq + 4 * g + e, w[e]);
                    } else {
                        *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                    }
                }
            if (g == 0) {
                const float lv = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_log2 + __log2f(l_tot));
                if (kLab && merge_mode == 2) store_dev(lpart + row, lv);
                else lpart[row] = lv;
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
                        if constexpr ((ABL & 1024) != 0) {      // LAB A/B: O written once — nt
                            typedef unsigned nt4 __attribute__((ext_vector_type(4)));
                            const nt4 t = {r0[0], r1[0], r0[1], r1[1]};
                            __builtin_nontemporal_store(t, (nt4*)(optr + 32 * db + 8 * (2 * pr + g)));
                        } else {
                            *(uint4*)(optr + 32 * db + 8 * (2 * pr + g)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        }
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
    // single-launch merge of the key-range shares (attn_common.h)
    if (kLab && nsplit > 1 && done != nullptr)
        prefill_release_and_merge<T, HD>(p, nsplit, b, h, q_wg0, min(Sq, q_wg0 + BM), q_first, done + ((int64_t)b * p.h + h) * nqb + qb, (int*)(smem + VBASE + 3 * S::kTileBytes), merge_mode);
}

// host side: grid as prefill_kernels.hip's 1-D / 3-D orders with 256-row query blocks
dim3 prefill_grid(const vattn_attn_params* p, int nqb, int* order_out);       // prefill_kernels.hip

constexpr int kSmem64 = 36864 + 3 * PfSmem<128>::kTileBytes;      // K ring (padded layout: 2 x 17 408, rounded up) + V ring
template <typename T, int ABL, int NA, int RING, int MS = 8, int BJ = 8, int D0 = 9, int DS = 3, int R6 = 0, int XTRA = 0> static void launch64_t(const vattn_attn_params* p, hipStream_t st, int nsplit, int* done, int merge_mode) {
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
        (void)hipFuncSetAttribute((const void*)prefill64_kernel<T, ABL, NA, RING, MS, BJ, D0, DS, R6, XTRA>, hipFuncAttributeMaxDynamicSharedMemorySize, (R6 & 2) ? kStampBase + 8192 : kSmem64 + 16);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((prefill64_kernel<T, ABL, NA, RING, MS, BJ, D0, DS, R6, XTRA>), grid, dim3(256), (R6 & 2) ? kStampBase + 8192 : kSmem64 + 16, st, *p, order, nqb, nsplit, done, merge_mode);
}

// Product: ONE build per dtype (padded K image, 24 exp2 pairs in phase A, fragment ring of 4, row-max chain in groups 8-23, barrier at group 8, DMA in groups 9, 12, .. 30).  The lab library (-DVATTN_LAB) adds the
// K-image alternative and the timing ablations of tools/kbench.py behind variant bits 8-11 (0 = product; 1-3, 11, 12, 14 = correct alternatives;
// 4-9 = ablations whose RESULTS ARE WRONG).
void launch_prefill64(const vattn_attn_params* p, hipStream_t st, int nsplit, int* done, int merge_mode) {
#ifdef VATTN_LAB
    // round 6 experiments, variant bits 28-30 (fp16): 1 = V^T fragments pre-read in phase A's tail; 3 = the same with stamps
    switch ((p->variant >> 28) & 7) {
        case 1: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1>(p, st, nsplit, done, merge_mode);
        case 3: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 3>(p, st, nsplit, done, merge_mode);
        // encoding-size probes (RESULTS ARE WRONG: the maximum is not subtracted): the v_fma as a 4-byte / an 8-byte v_mul
        case 2: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 16>(p, st, nsplit, done, merge_mode);
        case 4: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 32>(p, st, nsplit, done, merge_mode);
        // (call 2 counted the issue-budget ablations — no v_fma: ABL 2048, no v_add: 4096, ...: profiles/r06_p64_issue_ablation_pmc.txt)
        // (round-6 call 2 also counted: no v_fma + no v_add, no fma / exp / add, MFMAs + reads + DMA only: profiles/r06_p64_issue_ablation_pmc.txt)
        case 5: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4>(p, st, nsplit, done, merge_mode);             // scale in a scalar register
        // (call 3 also counted the stage order M, E, A inside a group: slower, profiles/r06_p64_scalar_scale_pmc.txt)
        case 6: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 64>(p, st, nsplit, done, merge_mode);            // v_mov + v_fmac (4-byte encodings), exact
        case 7:      // sub-selected by split_reserved bits 16-23 (tools/kbench.py: KBENCH_LAB_SUB)
            switch ((p->split_reserved >> 16) & 255) {
                case 0: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 64 | 128>(p, st, nsplit, done, merge_mode);      // + two v_max for each v_max3, exact
                // price list: the scalar-scale schedule + 64 extra instructions of ONE class per tile (results unchanged)
                case 1: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 1>(p, st, nsplit, done, merge_mode);
                case 2: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 2>(p, st, nsplit, done, merge_mode);
                case 3: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 3>(p, st, nsplit, done, merge_mode);
                case 4: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 4>(p, st, nsplit, done, merge_mode);
                case 5: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 5>(p, st, nsplit, done, merge_mode);
                case 6: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 6>(p, st, nsplit, done, merge_mode);
                case 7: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 7>(p, st, nsplit, done, merge_mode);
                case 8: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 8>(p, st, nsplit, done, merge_mode);
                case 9: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 9>(p, st, nsplit, done, merge_mode);
                case 10: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 10>(p, st, nsplit, done, merge_mode);
                case 11: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4, 11>(p, st, nsplit, done, merge_mode);
                case 12: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4 | 256>(p, st, nsplit, done, merge_mode);      // row sums on the matrix pipe (closed: slower, profiles/r06_p64_rowsum_mfma_pmc.txt)
                case 13: return launch64_t<_Float16, 128, 24, 4, 8, 8, 9, 3, 1 | 4 | 512>(p, st, nsplit, done, merge_mode);      // DMA piece distances in the scalar offset
                default: break;
            }
            break;
        default: break;
    }
    const int sel = (p->variant >> 8) & 15;
    if (p->dtype == VATTN_DTYPE_BF16) {
        if (sel == 3) return launch64_t<__bf16, 0, 24, 4>(p, st, nsplit, done, merge_mode);
    } else {
        switch (sel) {
            case 3: return launch64_t<_Float16, 0, 24, 4>(p, st, nsplit, done, merge_mode);                    // XOR-swizzled K image (round 2's first layout)
            // schedule alternatives (correct, bit-identical results; tools/p64_variants.py, profiles/r03_p64_schedules.txt)
            case 14: return launch64_t<_Float16, 128, 24, 4>(p, st, nsplit, done, merge_mode);               // the product build inside the lab library (lab vs
                                                                                                              // product binaries of one source differ by up to 1 %)
            case 1: return launch64_t<_Float16, 128, 20, 4, 8, 8, 9, 3>(p, st, nsplit, done, merge_mode);    // 20 exp2 pairs in phase A
            case 2: return launch64_t<_Float16, 128, 24, 4, 4, 8, 9, 3>(p, st, nsplit, done, merge_mode);    // row-max chain from group 4
            case 11: return launch64_t<_Float16, 128, 24, 4, 8, 12, 13, 2>(p, st, nsplit, done, merge_mode); // barrier after group 11, DMA every second group
            case 12: return launch64_t<_Float16, 128, 24, 4, 8, 24, 24, 1>(p, st, nsplit, done, merge_mode); // round 2's placement: barrier after 23, DMA 24-31
            case 15: return launch64_t<_Float16, 1024 | 128, 24, 4>(p, st, nsplit, done, merge_mode);         // A/B (round 4): non-temporal Q loads and O stores
            case 10: return launch64_t<_Float16, 256 | 128, 24, 4>(p, st, nsplit, done, merge_mode);          // ablation: no epilogue stores
            case 13: return launch64_t<_Float16, 512 | 128, 24, 4>(p, st, nsplit, done, merge_mode);          // ablation: no zero-fill of the V ring
            case 4: return launch64_t<_Float16, 1 | 128, 24, 4>(p, st, nsplit, done, merge_mode);              // no LDS-DMA in the steady state
            case 5: return launch64_t<_Float16, 2 | 128, 24, 4>(p, st, nsplit, done, merge_mode);              // no fma / exp2 / row sums
            case 6: return launch64_t<_Float16, 8 | 128, 24, 4>(p, st, nsplit, done, merge_mode);              // no per-tile wait + barrier
            case 7: return launch64_t<_Float16, 16 | 128, 24, 4>(p, st, nsplit, done, merge_mode);             // no LDS fragment reads
            case 8: return launch64_t<_Float16, 1 | 2 | 4 | 32 | 128, 24, 4>(p, st, nsplit, done, merge_mode);           // MFMAs + fragment reads + barrier
            case 9: return launch64_t<_Float16, 1 | 2 | 4 | 8 | 16 | 32 | 128, 24, 4>(p, st, nsplit, done, merge_mode);  // MFMAs only
            default: break;
        }
    }
#endif
    if (p->dtype == VATTN_DTYPE_BF16) launch64_t<__bf16, 128, 24, 4>(p, st, nsplit, done, merge_mode);
    else launch64_t<_Float16, 128, 24, 4>(p, st, nsplit, done, merge_mode);
}

}  // namespace vattn_k
