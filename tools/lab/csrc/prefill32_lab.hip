// LAB (round 6; tools/lab/libvattn_lab.so, explicit tiling 3 = variant 6): the data flow of prefill64_kernel — K / V tiles by LDS-DMA into a
// 2 / 3 slot ring shaped on the global side (padded K pieces, V^T sub-tiles for ds_read_b64_tr_b16), one barrier per tile with hand-counted
// vmcnt, the swapped product S^T = K.Q^T with the softmax entirely in registers, the in-wave software pipeline
//   phase A  S(t+1) = K(t+1).Q^T  ||  P(t) = exp2(S(t)), row sums      phase B  O += V(t)^T.P(t)^T  ||  row max of S(t+1), f16 packing of P(t)
// with hand-placed groups {MFMA ; fragment read ahead ; a slice of softmax VALU}, deferred rescale, running descriptors in fixed SGPR quads —
// on EIGHT waves of 32 query rows: TWO WAVES PER SIMD, 256 registers each, one 256-row query block per 512-thread workgroup.
// Why it was built: prefill64_kernel (4 waves x 64 rows, one wave per SIMD) is bound by what ONE wave can issue to the vector ALU (7.3 cycles per
// VALU instruction beside the MFMAs, the matrix pipe idle a third of the time: profiles/r06_p64_price_list.txt), and the bare instruction streams
// said a second wave would help (tools/lab/issue_probe3.cpp, profiles/r06_issue_probe3.txt: x 1.14-1.16 as two 32-row waves).
// RESULT (profiles/r06_prefill32_ab.txt, r06_prefill32_pmc.txt): parity-green on the first run, and it does take FEWER CYCLES — 2 912 against 3 031
// chip cycles per 64 MFMAs per SIMD on the configs[1] prompt, matrix-pipe duty 0.703 against 0.676 — but MORE TIME (7.90 against 7.69-7.76 ms,
// + 2-3 % on every chip-filling shape): the part is POWER-bound in this kernel and clocks the 32-row form 6 % lower (1 525 against 1 616 MHz in
// the counter pass).  A 32-row wave's K / V^T fragment feeds ONE MFMA instead of two: LDS reads per MFMA double (1.50 against 0.75
// instructions, the LDS 35 % busy against 17 %), VALU per MFMA 4.75 against 4.38 — energy that the cycles saved do not pay for.  Six
// placements of the barrier / DMA / softmax split / ring depth (tools/lab/p32_builds.sh) sit within +- 0.5 % of each other.  Closed: the plan
// keeps prefill64_kernel; what would help is fewer joules per tile, not fewer cycles.
// Softmax exactly as the reference states it (softmax.h:69-94; see prefill64_kernels.hip for the numerics notes).  Semantics:
// /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, mask.h:164-196 (bottom-right causal), softmax.h:69-157 (fp32 max / sum
// via exp2, P rounded to the I/O dtype before PV), flash_fwd_kernel.h:57-499; :1116-1297 (split combine = combine_rows_kernel / combine_blocks_kernel).
// Every K/V access is bounded by a buffer descriptor that ends at the sequence's visible length.
#include "prefill64_common.h"

namespace vattn_k {

// The MFMAs of this kernel name no accumulator-half registers: with 256 registers per wave (two waves per SIMD) the compiler's fixed
// 128 / 128 split between the architectural and the accumulator half leaves the tile step 14 registers short on the architectural side
// (it then copies through v_accvgpr_write / read inside the step); a kernel whose inline asm has no "a" operand gets ONE file of 256.
template <typename T> struct Mfma32;
#define VATTN_MFMA32_STRUCT(TYPE, MNEM)                                                                                              \
    template <> struct Mfma32<TYPE> {                                                                                              \
        using V8 = typename Tr<TYPE>::v8;                                                                                          \
        static __device__ __forceinline__ void qk_first(f32x16& d, V8 a, V8 b) { asm volatile(MNEM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); } \
        static __device__ __forceinline__ void qk_acc(f32x16& d, V8 a, V8 b) { asm volatile(MNEM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }   \
        static __device__ __forceinline__ void pv(f32x16& o, V8 a, V8 b) { asm volatile(MNEM " %0, %1, %2, %0" : "+v"(o) : "v"(a), "v"(b)); }       \
    };
VATTN_MFMA32_STRUCT(_Float16, "v_mfma_f32_32x32x16_f16")
VATTN_MFMA32_STRUCT(__bf16, "v_mfma_f32_32x32x16_bf16")
#undef VATTN_MFMA32_STRUCT

// The schedule of a wave's tile step (32 MFMA groups: 16 of phase A, 16 of phase B): 12 of the tile's 16 exp2 pairs start in phase A, a fragment
// ring of RING (RING - 1 fragments = MFMAs ahead), the row-max chains in phase-B groups 4-11, the per-tile wait + barrier in front of
// group BJ, one LDS-DMA piece in groups D0, D0 + DS, ... (a wave fetches two K pieces and two V pieces of the tile's 32).
template <typename T>
__global__ __launch_bounds__(512, 1) void prefill32_kernel(vattn_attn_params p, int order, int nqb, int nsplit) {
#ifndef P32_SCHED      // (tools/lab/p32_builds.sh builds other placements: -DP32_SCHED=NA,RING,MS,BJ,D0,DS)
#define P32_SCHED 12, 4, 4, 4, 5, 3
#endif
    constexpr int kSched[6] = {P32_SCHED};
    constexpr int NA = kSched[0], RING = kSched[1], MS = kSched[2], BJ = kSched[3], D0 = kSched[4], DS = kSched[5];
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
    static_assert(D0 >= BJ && D0 + 3 * DS < 16, "DMA pieces behind the barrier, inside phase B");
    static_assert(NA >= 8 && NA < 16, "key slice 0 of P is packed in phase-A group 7: its four pairs must be exponentiated by group 6");
    static_assert(MS >= 4 && MS + 11 < 16, "row-max chains >= 4 MFMAs behind the last S^T MFMA, their reduction inside phase B");
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
    const int qw0 = q_wg0 + wave * 32;                         // first query row of this wave

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
    // a wave fetches pieces pc = 2*wave + j, j = 0, 1, of the K tile's 16 and of the V tile's 16:
    // K piece pc holds rows 4*pc .. 4*pc+3: lane i -> row 4*pc + (i & 3), 16-byte chunk i >> 2 of that row
    // V piece pc = (d block pc >> 2, keys 16*(pc & 3) .. +15): lane i -> key 16*(pc & 3) + (i >> 2), global chunk 4*(pc >> 2) + (i & 3)
    unsigned koff[2], voff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int pc = 2 * wave + j;
        const int row = 4 * pc + (lane & 3);
        koff[j] = (unsigned)row * k_rs_bytes + (unsigned)((lane >> 2) << 4);
        const int key = 16 * (pc & 3) + (lane >> 2);
        voff[j] = (unsigned)key * v_rs_bytes + (unsigned)((4 * (pc >> 2) + (lane & 3)) << 4);
    }
    using M = Mfma32<T>;
    const unsigned k_lds_wave = (unsigned)(wave * 2 * KPIECE);                 // this wave's two K pieces inside a K slot
    const unsigned v_lds_wave = (unsigned)(VBASE + wave * 2048);               // ... and V pieces inside a V slot (piece pc at pc * 1024)
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
    auto dma_k_all = [&](int t) {      // K(t) -> K slot (t - tb) & 1, this wave's two pieces
        const u32x4 r = k_rsrc(t);
        const unsigned l0 = k_lds_wave + (unsigned)(kslot(t) * KSLOT);
        dma_piece_first(l0, r, koff[0]);
        dma_piece(l0 + KPIECE, r, koff[1]);
    };
    auto dma_v_all = [&](int t) {
        const u32x4 r = v_rsrc(t);
        const unsigned l0 = v_lds_wave + (unsigned)(((t - tb) % 3) * S::kTileBytes);      // prologue only: V(tb) -> slot 0, V(tb+1) -> slot 1
        dma_piece_first(l0, r, voff[0]);
        dma_piece(l0 + 1024, r, voff[1]);
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
        for (int i = 0; i < (3 * S::kTileBytes) / (512 * 16); i++) *(uint4*)(smem + VBASE + (i * 512 + tid) * 16) = z;
        __syncthreads();
    }
    // (asking for V(tb) and K(tb+1) only once Q sits in its registers — so that the wait for Q does not also wait for them — was measured:
    // short pieces lose more on the later K(tb+1) than the first S' gains)
    dma_k_all(tb);
    dma_v_all(tb);
    dma_k_all(tb + 1);

    // Q^T fragments (B operand of S^T = K.Q^T): slot (g, j) <-> d = 16*kk + 8*g + j; pre-scaled into the log2 domain
    const float escale = p.softmax_scale * kLog2e;                      // raw score -> log2 domain
    V8 qf[KK];
    {
        const int my_q = qw0 + l31;
        const T* qptr = (const T*)p.q + (p.q_start ? 0 : (int64_t)b * p.q_batch_stride) + (q_first + my_q) * p.q_row_stride + (int64_t)h * p.q_head_stride;
        V8 raw[KK];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (my_q < Sq) v = *(const uint4*)(qptr + 16 * kk + 8 * g);
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
            qf[kk] = raw[kk];
        }
    }

    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; i++) o[i] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // -(running max) * softmax_scale * log2e of the lane's query: the addend of the exp2 argument (softmax.h:86-94)
    float nmsub = 0.f;
    // lane-local partial row sums (the other half-lane holds the other 32 keys of every tile), two independent accumulators
    float l_acc[2] = {0.f, 0.f};

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
    auto mask_tile = [&](int tt, f32x16 (&s)[2]) {
        const int n0 = tt * PF_BN;
        const int my_q = qw0 + l31;
        const int lim = causal ? min(Lk - 1, my_q + off) : Lk - 1;     // last visible key of this query
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = n0 + 32 * kb + 8 * (r >> 2) + 4 * g + (r & 3);
                if (key > lim) s[kb][r] = -INFINITY;
            }
    };
    // tile tt needs masking (ragged end of the sequence / causal diagonal of this wave's rows) iff tt >= t_mask:
    // 64 tt + 64 > Lk  <=>  tt >= Lk >> 6;   64 tt + 63 > qw0 + off  <=>  tt >= ((qw0 + off - 63) >> 6) + 1 (arithmetic shift)
    const int t_mask = min(Lk >> 6, causal ? ((qw0 + off - 63) >> 6) + 1 : 0x7fffffff);
    auto needs_mask = [&](int tt) -> bool { return tt >= t_mask; };
    auto row_max = [&](const f32x16 (&s)[2]) -> float {
        float m0 = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; r++) m0 = fmaxf(fmaxf(m0, s[0][r]), s[1][r]);     // v_max3_f32
        return fmaxf(m0, swap_halves(m0));
    };
    // moves the running maximum up by delta >= 0 (log2 units, per lane): everything accumulated at the old scale — O and l — is rescaled
    // exactly once (cdna guide T13); scores not yet exponentiated are raw and take the new maximum
    auto raise_max = [&](float delta) {
        const float alpha = fast_exp2(-delta);
        nmsub -= delta;
#pragma unroll
        for (int i = 0; i < DB; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= alpha;
        l_acc[0] *= alpha;
        l_acc[1] *= alpha;
    };
    // P(t) -> the PV B-operand fragment of key slice ks: slot (g, j) <-> P registers 8*(ks&1) + j of key block ks>>1
    auto pack_p = [&](const f32x16 (&pt)[2], int ks) -> V8 {
        V8 r;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = X::cvt(pt[ks >> 1][8 * (ks & 1) + j]);
        return r;
    };

    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // this wave's pieces of K(tb) landed; V(tb), K(tb+1) may still fly
    __builtin_amdgcn_s_barrier();

    f32x16 sc[2];      // S(t): raw scores of the current tile (two 32-key blocks); becomes P(t) in place
    f32x16 sd[2];
    {
        const char* ksm = smem + kslot(tb) * KSLOT;
#pragma unroll
        for (int f = 0; f < 2 * KK; f++) {
            const V8 a = kfrag(ksm, f);
            if (f < 2) M::qk_first(sc[f & 1], a, qf[f >> 1]);
            else M::qk_acc(sc[f & 1], a, qf[f >> 1]);
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // the last MFMA results are VALU-readable from here
        SCHED_FENCE();
        if (needs_mask(tb)) mask_tile(tb, sc);
        const float mx = row_max(sc);
        nmsub = (mx == -INFINITY) ? 0.f : -mx * escale;     // softmax.h: a fully masked row keeps a zero reference
    }

    // ---- software pipeline of the softmax VALU work, in units of PAIRS of scores (pair e of the wave's 16: key block e>>3, half (e>>2)&1,
    // registers 8*half + 2*(e&3), +1 — the order in which the P.V key slices consume them: slice ks takes pairs 4 ks .. 4 ks + 3).  Stage E (two
    // v_exp) of pair e sits in group GE(e) of the step's 32 MFMA groups, stage M (two v_fma: s*scale*log2e - m*scale*log2e) one group
    // earlier, stage A (two v_add into the two row-sum accumulators) one group later.
    auto GE = [](int e) { return e < NA ? 1 + (e * 13) / NA : 17 + ((e - NA) * 9) / (16 - NA); };
#define P32_X0(cur, e) cur[(e) >> 3][8 * (((e) >> 2) & 1) + 2 * ((e) & 3)]
#define P32_X1(cur, e) cur[(e) >> 3][8 * (((e) >> 2) & 1) + 2 * ((e) & 3) + 1]
    const unsigned escale_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, escale));      // a REAL scalar register (prefill64_kernels.hip)
    auto softmax_stages = [&](int G, f32x16 (&cur)[2]) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if (GE(e) - 1 == G)
                asm("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(P32_X0(cur, e)), "+v"(P32_X1(cur, e)) : "s"(escale_s), "v"(nmsub));
            if (GE(e) == G) asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(P32_X0(cur, e)), "+v"(P32_X1(cur, e)));
            if (GE(e) + 1 == G)
                asm("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(l_acc[0]), "+v"(l_acc[1]) : "v"(P32_X0(cur, e)), "v"(P32_X1(cur, e)));
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
    auto step = [&](int t, const int par, f32x16 (&cur)[2], f32x16 (&nxt)[2], V8& kf0, V8& kf1, V8& kf2) {
        // par = (t - tb) & 1, a literal at both call sites: with the padded K layout every K fragment address folds to lane + immediate
        const int s_cur = par;                                                  // slot of K(t), K(t+2)
        const char* ksm = smem + (s_cur ^ 1) * KSLOT;                           // K(t+1)
        const char* ksm_next = smem + s_cur * KSLOT;                            // K(t+2)
        const char* vsm = smem + VBASE + vs_cur;                                // V(t)
        const unsigned lk0 = k_lds_wave + (unsigned)((s_cur ^ 1) * KSLOT);      // K(t+3) -> the slot K(t+1) leaves
        unsigned lv0 = 0;                                                       // V(t+2)'s pieces of this wave (set in phase A)
        const bool mask_next = needs_mask(t + 1);                               // ragged end / causal diagonal: wave-uniform, the last tiles only
        // Every wave runs the SAME straight-line body for every tile of the workgroup (see prefill64_kernels.hip).
        // ---------------- 32 groups of { MFMA ; fragment read ahead ; a slice of softmax VALU } ----------------
        // phase A: S'(t+1) = K(t+1).Q^T   (16 MFMAs: fragment f = i: k-step kk = i>>1, key block i&1)
        V8 pf[2];            // P(t) fragments of the key slice being multiplied and of the next one
        V8 kf[RING];         // RING - 1 fragments (as many MFMAs) ahead of their use
        V8 vf[RING];         // V(t)^T fragments of phase B
        SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int f = i;
            if (f < 3) {
                const V8 a = f == 0 ? kf0 : (f == 1 ? kf1 : kf2);
                if (i < 2) M::qk_first(nxt[f & 1], a, qf[f >> 1]);
                else M::qk_acc(nxt[f & 1], a, qf[f >> 1]);
            } else M::qk_acc(nxt[f & 1], kf[f % RING], qf[f >> 1]);
            // (fragments 0-2 came from the previous step; fragment f + RING - 1 is asked for here: 3 .. 15)
            if (f + RING - 1 >= 3 && f + RING - 1 < 2 * KK) kf[(f + RING - 1) % RING] = kfrag(ksm, f + RING - 1);
            if (i == 0) {
#pragma unroll
                for (int ff = 3; ff < RING - 1; ff++) kf[ff % RING] = kfrag(ksm, ff);      // (RING > 4: the ring's head)
            }
            softmax_stages(i, cur);
            // key slice 0 of P(t) (pairs 0-3: exponentiated by group GE(3) <= 6) is packed HERE, so the first P.V MFMA does not wait for it
            if (i == 7) pf[0] = pack_p(cur, 0);
            // the DMA stream's scalars move one tile on (SALU work, inside gaps)
            if (i == 8) lv0 = v_lds_wave + vs_dma;
            if (i == 9) asm volatile("s_mov_b32 %1, %0\n\ts_add_u32 %0, %0, %2\n\ts_cmp_eq_u32 %0, %3\n\ts_cselect_b32 %0, 0, %0"
                                     : "+s"(vs_cur), "=&s"(vs_dma) : "i"(S::kTileBytes), "i"(3 * S::kTileBytes) : "scc");
            if (i == 10) k_rsrc_advance(rk, k_rows_left, k_tile_b, k_rs_bytes);
            if (i == 11) v_rsrc_advance(rv, v_rows_left, v_tile_b, v_rs_bytes);
            // V(t) landed a step ago: its first fragments are asked for while the last S' MFMAs run
#pragma unroll
            for (int ff = 0; ff < RING - 1; ff++)
                if (i == 16 - (RING - 1) + ff) vf[ff] = vfrag(vsm, ff);
            SCHED_FENCE();
        }
        // phase B: O^T += V(t)^T.P(t)^T   (16 MFMAs: fragment f = j: key slice ks = j>>2, d block j&3)
        float mxa = -INFINITY, mxb = -INFINITY, mx = -INFINITY, g0 = -INFINITY;
        SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int f = j, ks = j >> 2;
            if (j == BJ) {
                // this wave's pieces of K(t+2) and V(t+1) (issued one step ago) have landed; behind the barrier everyone's have, and
                // every wave is past its reads of K(t+1) and V(t-1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // (a P fragment is packed at least one MFMA group before its first use: no VALU -> MFMA operand hazard to pad)
            M::pv(o[f & 3], vf[f % RING], pf[ks & 1]);
            if (f + RING - 1 < 16) vf[(f + RING - 1) % RING] = vfrag(vsm, f + RING - 1);
            softmax_stages(16 + j, cur);
            // the P fragment of key slice ks+1 is packed while slice ks is multiplied (4 cvt_pk, group 2 of a slice)
            if (ks < 3 && (j & 3) == 2) pf[(ks + 1) & 1] = pack_p(cur, ks + 1);
            // row max of S'(t+1): TWO chains of 8 (even / odd registers), groups MS .. MS+7 (>= 4 MFMAs after the last S^T MFMA was issued);
            // their join, the half-wave exchange and the growth test follow in the next gaps
            if (j >= MS && j < MS + 8) {
                const int r = 2 * (j - MS);
                if (r == 0) {      // the chains' first links need no -inf to start from
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mxa) : "v"(nxt[0][0]), "v"(nxt[1][0]));
                    asm("v_max_f32_e32 %0, %1, %2" : "=v"(mxb) : "v"(nxt[0][1]), "v"(nxt[1][1]));
                } else {
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mxa) : "v"(nxt[0][r]), "v"(nxt[1][r]));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mxb) : "v"(nxt[0][r + 1]), "v"(nxt[1][r + 1]));
                }
            }
            if (j == MS + 8) asm("v_max_f32_e32 %0, %1, %2" : "=v"(mx) : "v"(mxa), "v"(mxb));
            if (j == MS + 9) mx = max_halves(mx);
            // growth of the row maximum over the running maximum, log2 units (nmsub = -m*scale*log2e; -inf for rows that see nothing here)
            if (j == MS + 10) asm("v_fma_f32 %0, %1, %2, %3" : "=&v"(g0) : "v"(mx), "s"(escale_s), "v"(nmsub));
            // this wave's two pieces of K(t+3) and of V(t+2): the second takes the first's per-lane offset, its distance (4 K rows | 16 V keys)
            // travels in the load's scalar offset (prefill64_common.h)
            if (j == dma_gap(0)) dma_piece_at<0>(lk0, rk, koff[0]);
            if (j == dma_gap(1)) dma_piece_so<KPIECE, 4>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(2)) dma_piece_at<0>(lv0, rv, voff[0]);
            if (j == dma_gap(3)) dma_piece_so<1024, 16>(lv0, rv, voff[0], v_rs_bytes);
            if (j == 12) kf0 = kfrag(ksm_next, 0);          // the next step's first K fragments: K(t+2) is behind the barrier
            if (j == 13) kf1 = kfrag(ksm_next, 1);
            if (j == 14) kf2 = kfrag(ksm_next, 2);
            SCHED_FENCE();
        }
        if (mask_next) {
            mask_tile(t + 1, nxt);
            mx = row_max(nxt);
            g0 = __builtin_fmaf(mx, escale, nmsub);
        }
        if (__builtin_amdgcn_ballot_w64(g0 > kDeferLog2) != 0) {            // rare: a row's maximum grew by > 2^6
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");                        // every PV result has landed in O
            SCHED_FENCE();
            raise_max(fmaxf(g0, 0.f));
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
#undef P32_X0
#undef P32_X1

    // ---- epilogue: O^T[d = 32*db + 8*(r>>2) + 4*g + (r&3)][query] ----
    // (Staging the fp32 partials of a key-range piece through LDS so that every store instruction writes whole 512-byte rows instead of
    // 32 bytes of 32 rows was built and measured in round 3: no gain — the cost of the partials (no-store ablation: 5-19 % of the
    // tensor-parallel launches, profiles/r03_p64_prologue_epilogue.txt) is their volume, not their coalescing.)
    {
        const int my_q = qw0 + l31;
        const float l_loc = l_acc[0] + l_acc[1];
        const float l_tot = l_loc + swap_halves(l_loc);
        const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
        const float m_log2 = -nmsub;                          // running max of softmax_scale*log2e*q.k
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
                    for (int e = 0; e < 4; e++) w[e] = o[db][4 * tq + e] * inv;
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
                            we[e] = X::cvt(o[db][4 * (2 * pr) + e] * inv);
                            wo[e] = X::cvt(o[db][4 * (2 * pr + 1) + e] * inv);
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
                        for (int e = 0; e < 4; e++) w[e] = X::cvt(o[db][4 * tq + e] * inv);
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
}

// host side: grid as prefill_kernels.hip's 1-D / 3-D orders with 256-row query blocks (the same blocks, lists and splits as prefill64_kernel)
dim3 prefill_grid(const vattn_attn_params* p, int nqb, int* order_out);       // prefill_kernels.hip

constexpr int kSmem32 = 36864 + 3 * PfSmem<128>::kTileBytes;      // K ring (2 x 17 408, rounded up) + V ring
template <typename T> static void launch32_t(const vattn_attn_params* p, hipStream_t st, int nsplit) {
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
        (void)hipFuncSetAttribute((const void*)prefill32_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem32 + 16);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((prefill32_kernel<T>), grid, dim3(512), kSmem32 + 16, st, *p, order, nqb, nsplit);
}

void launch_prefill32(const vattn_attn_params* p, hipStream_t st, int nsplit) {
    if (p->dtype == VATTN_DTYPE_BF16) launch32_t<__bf16>(p, st, nsplit);
    else launch32_t<_Float16>(p, st, nsplit);
}

}  // namespace vattn_k
