// Persistent form of prefill64_kernel for WORK-LIST launches (vattn_prefill_plan_wg): one workgroup per CU walks a host-assigned QUEUE
// of (entry, head, 256-row query block, key-tile range) pieces, and the tile stream does not stop between them (round 5; DESIGN §5,
// VERDICT r04 next-round item 1; the operator: /root/reference/pod_attn/pod_attn/flash_fwd_kernel.h:57-499, its launch rule
// flash_fwd_launch_template.h:60-61,239-263 — one CTA per query block, which is what prefill64_kernel also does).
//
// What a piece costs in prefill64_kernel besides its key tiles: the workgroup's dispatch, two levels of dependent scalar loads (the
// piece, then the lengths it points to), a cold prologue — Q from HBM, the first K / V tiles' round trips, the first S' with the matrix
// pipe otherwise idle — and the drain of the DMA ring; the planner prices it at 3 tile times (csrc/prefill_kernels.hip), which is
// 3 % of the dynamic legs' average piece (99 tiles) and 10-25 % of the cut pieces of a tensor-parallel shard (12-30 tiles).  Here:
//   * the DMA stream is CONTINUOUS across pieces: the K(t+3) / V(t+2) fetches issued in the last three steps of a piece are the next
//     piece's first tiles (the running descriptors are re-based instead of advanced: K in step nt-3, V in step nt-2);
//   * the next piece's Q block travels HBM -> LDS by LDS-DMA while the current piece computes (a per-wave 16 KiB staging area behind the
//     V ring) and is moved into the accumulator registers between the last two steps; the LAST step of a piece then computes
//     S'(first tile of the next piece) in its phase A, exactly as every other step computes S'(t+1);
//   * between two pieces only the epilogue of the finished one is left: normalise and store O (or the fp32 partial), zero the
//     accumulators, take the new piece's first row maxima as its reference.
// A piece shorter than 3 tiles does not chain OUT (its re-base points would lie in its predecessor's steps); the piece after it starts
// cold — the same prologue as prefill64_kernel's.  Fused RoPE is not taken here (the launch keeps prefill64_kernel for it).
// The tile step itself — the 64 hand-placed groups { MFMA ; fragment read ahead ; softmax slice } — is prefill64_kernel's product
// instantiation (padded K image, NA = 24, fragment ring of 4, barrier at group 8, DMA in groups 9, 12, .. 30), restated here with
// the four hooks the stream needs: descriptor re-base, mask parameters of the tile being scored, exported row maxima, no max-growth
// test across a piece boundary.
#include "prefill64_common.h"

namespace vattn_k {

constexpr int kQStage = 87040;                 // LDS offset of the Q staging area: 4 waves x 16 KiB (K ring 36 864 + V ring 49 152 + 16, rounded up to 1 KiB)
constexpr int kSmem64p = kQStage + 65536;      // 152 576 bytes: one workgroup per CU (160 KiB of LDS)

// DYN = false: workgroup w walks its own queue, pieces [pf_wg_first[w], pf_wg_first[w + 1]) of a list GROUPED by workgroup (the host assigned
// them: vattn_prefill_plan_wg).  DYN = true: the list is in longest-first order and the workgroups DRAW their pieces — workgroup w starts
// with piece w and takes further pieces of its XCD's sub-sequence (positions = w mod 8, when the grid is a multiple of 8: an XCD's L2 keeps
// seeing the kv heads the list order gives it) from a counter in device memory, one ticket ahead of need: a static assignment cannot
// absorb the few per cent by which workgroups differ in speed (profiles/r05_p64p_legs_ab_static_queues.txt: 6-8 % slower than the
// hardware's own greedy dispatch of one workgroup per piece on the ragged tensor-parallel batches), a drawn one does what the dispatcher
// does.  `ctr`: 8 zero-initialised counters owned by the library (one set per device and stream); the last draw of a queue resets it.
template <typename T, bool DYN, int NA, int RING, int MS = 8, int BJ = 8, int D0 = 9, int DS = 3>
__global__ __launch_bounds__(256, 1) void prefill64p_kernel(vattn_attn_params p, int* ctr) {
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int HD = 128;
    using S = PfSmem<HD>;
    constexpr int BM = 256;
    constexpr int KK = HD / 16;
    constexpr int DB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K ring, V ring (prefill64_kernel's map), 16 spare bytes, Q staging; LDS address 0
    constexpr int KPIECE = 1088;
    constexpr int KSLOT = 16 * 1088;
    constexpr int VBASE = 36864;
    static_assert(D0 >= BJ && D0 + 7 * DS < 32, "DMA pieces behind the barrier, inside phase B");
    static_assert(NA >= 16 && NA < 32, "key slice 0 of P is packed in phase-A groups 13 / 15");
    static_assert(MS >= 4 && MS + 19 < 32, "row-max chain inside phase B");
    auto dma_gap = [](int k) { return D0 + DS * k; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const bool causal = p.is_causal != 0;
    const int G = p.h / p.h_k;
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;

    // ---- this workgroup's queue ----
    // static: pieces [q_idx, q_end) of the grouped list, in order.  drawn: piece blockIdx.x first, then positions qx + nq * (gq + ticket)
    int q_idx, q_end = 0;
    int next_idx = -1;                                    // drawn: the piece after `cur` (-1: none)
    const int n_items = p.num_pf_items;
    const int nq = (gridDim.x & 7) == 0 ? 8 : 1, qx = (int)blockIdx.x % nq, gq = (int)gridDim.x / nq;
    const int n_q = (n_items - qx + nq - 1) / nq;        // positions of this queue; its tickets are 0 .. n_q - 1 (gq of them draw nothing)
    int* const lds_word = (int*)(smem + 86016);           // the 16 spare bytes behind the V ring: wave 0 hands the drawn piece to the others
    if (DYN) {
        q_idx = (int)blockIdx.x;
        if (q_idx >= n_items) return;
    } else {
        q_idx = __builtin_amdgcn_readfirstlane(p.pf_wg_first[blockIdx.x]);
        q_end = __builtin_amdgcn_readfirstlane(p.pf_wg_first[blockIdx.x + 1]);
        if (q_idx >= q_end) return;
    }
    // One draw from this queue's counter by lane 0 of the calling wave (wave 0 calls it).  The old value arrives in v255 a memory round trip
    // LATER — behind a vmcnt wait the caller provides (a tile step's) — so it must not be an output of the statement: the compiler would
    // take the register's content at the statement for the value and might copy it away (park it in the accumulator file across the
    // step) before it has arrived.  v255 is named only here and in publish(); the allocator hands registers out from v0 upward and this
    // kernel needs fewer than 250: vattention_amd/build.py checks on the generated assembly that nothing else touches it.
    auto draw = [&]() {
        unsigned long long save;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add v255, %1, %2, %3 sc0\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "v"((unsigned)(qx * 4)), "v"(1u), "s"(ctr) : "memory", "v255");
    };
    // the ticket that arrived -> the piece it stands for (or -1), into the LDS word; the queue's last ticket zeroes the counter for the next launch
    auto publish = [&]() {
        int tv;
        asm volatile("v_readfirstlane_b32 %0, v255" : "=s"(tv) : : "memory");
        const int pos = qx + nq * (gq + tv);
        if (tv == n_q - 1 && lane == 0) ctr[qx] = 0;
        if (lane == 0) *lds_word = pos < n_items ? pos : -1;
    };

    // One piece, resolved against the DEVICE-side lengths (the list is a hint: include/vattn_kernels.h).  Its fifteen scalars live in the
    // LANES of one vector register (field i in lane i; v_writelane / v_readlane): the kernel sits at the SGPR limit, the steady state
    // needs four of them (kept in scalars below), and the compiler's own spilling of thirty more put lane reads — and scratch reloads
    // with their vmcnt(0) — into the tile step.
    enum { F_B, F_H, F_QWG0, F_SQ, F_LK, F_OFF, F_TB, F_NT, F_ROW, F_QF_LO, F_QF_HI, F_KB_LO, F_KB_HI, F_VB_LO, F_VB_HI };
#define PF(rec, i) ((int)__builtin_amdgcn_readlane((int)(rec), (i)))
#define PF64(rec, i) (((unsigned long long)(unsigned)PF(rec, (i) + 1) << 32) | (unsigned)PF(rec, i))
    // (The rare blocks read the kernel arguments through a laundered pointer: the forty dwords they need — strides, table pointers — are
    // then loaded where they are used instead of once in front of the loop, where they would sit in SGPRs, be spilled to vector lanes
    // and push the steady state's vector registers out.)
    auto args = [&]() -> const vattn_attn_params& {
        // the parameter block is the kernel's only argument: offset 0 of the kernarg segment (taking &p would copy it to scratch)
        const vattn_attn_params* q = (const vattn_attn_params*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        return *q;
    };
    auto load_piece = [&](int idx) -> int {
        const vattn_attn_params& p = args();
        const vattn_prefill_item it = p.pf_items[idx];
        const int b = __builtin_amdgcn_readfirstlane(it.b), h = __builtin_amdgcn_readfirstlane(it.h);
        const int qb = __builtin_amdgcn_readfirstlane(it.qb);
        const int it_tb = __builtin_amdgcn_readfirstlane(it.tile_begin), it_te = __builtin_amdgcn_readfirstlane(it.tile_end);
        const int it_row = __builtin_amdgcn_readfirstlane(it.nshares > 1 ? it.part_row : -1);
        const int hk = h / G;
        const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
        int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
        Lk = Lk > p.seqlen_k ? p.seqlen_k : Lk;
        const int Sq = p.q_lens ? __builtin_amdgcn_readfirstlane(p.q_lens[b]) : p.seqlen_q;
        const long long q_first = p.q_start ? (long long)__builtin_amdgcn_readfirstlane(p.q_start[b]) : 0;
        const int off = Lk - Sq, q_wg0 = qb * BM;
        int n_end = Lk;
        if (causal) n_end = min(Lk, q_wg0 + BM + off);
        if (n_end < 0) n_end = 0;
        int nt_all = (n_end + PF_BN - 1) / PF_BN;
        if (q_wg0 >= Sq) nt_all = 0;                           // a block beyond its entry's rows: nothing to score, nothing to store
        const unsigned long long kb = (unsigned long long)((const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride);
        const unsigned long long vb = (unsigned long long)((const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride);
        // (v_writelane_b32 by inline asm: this compiler has the readlane builtin only)
        auto wl = [](int val, int ln, int rec) -> int {
            const int sv = __builtin_amdgcn_readfirstlane(val);      // (a value the compiler holds in a vector register would be printed as one)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(rec) : "s"(sv), "n"(ln));
            return rec;
        };
        int r = 0;
        r = wl(b, F_B, r);
        r = wl(h, F_H, r);
        r = wl(q_wg0, F_QWG0, r);
        r = wl(Sq, F_SQ, r);
        r = wl(Lk, F_LK, r);
        r = wl(off, F_OFF, r);
        r = wl(min(nt_all, it_tb), F_TB, r);
        r = wl(min(nt_all, it_te), F_NT, r);
        r = wl(it_row, F_ROW, r);
        r = wl((int)(unsigned)q_first, F_QF_LO, r);
        r = wl((int)(unsigned)((unsigned long long)q_first >> 32), F_QF_HI, r);
        r = wl((int)(unsigned)kb, F_KB_LO, r);
        r = wl((int)(unsigned)(kb >> 32), F_KB_HI, r);
        r = wl((int)(unsigned)vb, F_VB_LO, r);
        r = wl((int)(unsigned)(vb >> 32), F_VB_HI, r);
        return r;
    };
    // first tile of a wave's rows that needs masking (ragged end of the sequence / causal diagonal), prefill64_kernel's rule
    auto t_mask_of = [&](int rec) -> int { return min(PF(rec, F_LK) >> 6, causal ? ((PF(rec, F_QWG0) + wave * 64 + PF(rec, F_OFF) - 63) >> 6) + 1 : 0x7fffffff); };

    // (The rare blocks of the persistent loop — masking, epilogue, cold start — take their lane ids through an empty asm: what they derive
    // from them is then not loop-invariant, and LICM cannot hoist dozens of per-lane constants in front of the loop, where they would be
    // spilled to scratch and reloaded — with a compiler-placed vmcnt(0) — inside the tile steps.)
    auto opaque = [](int x) -> int {
        asm volatile("" : "+v"(x));
        return x;
    };

    // ---- DMA addressing (tile-invariant per-lane offsets; prefill64_kernel's padded K image and V sub-tiles) ----
    // K piece pc = 4*wave + j holds rows 4*pc .. 4*pc+3 (lane i -> row 4*pc + (i & 3), chunk i >> 2); V piece j = (d block wave, keys 16*j ..):
    // lane i -> key 16*j + (i >> 2), global chunk 4*wave + (i & 3).  Piece j's offset = piece 0's + j x (4 K rows | 16 V keys).
    unsigned koff[1], voff[1];
    koff[0] = (unsigned)(16 * wave + (lane & 3)) * k_rs_bytes + (unsigned)((lane >> 2) << 4);
    voff[0] = (unsigned)(lane >> 2) * v_rs_bytes + (unsigned)((4 * wave + (lane & 3)) << 4);
    using M = Mfma<T>;
    const unsigned k_lds_wave = (unsigned)(wave * 4 * KPIECE);
    const unsigned v_lds_wave = (unsigned)(VBASE + wave * 4096);
    const unsigned q_lds_wave = (unsigned)(kQStage + wave * 16384);
    auto tile_desc = [&](unsigned long long base, int t, int Lk_, unsigned rs_bytes) -> u32x4 {
        int rem = Lk_ - t * PF_BN;
        rem = rem < 0 ? 0 : (rem > PF_BN ? PF_BN : rem);
        const unsigned long long a = base + (unsigned long long)t * PF_BN * rs_bytes;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
        r[2] = __builtin_amdgcn_readfirstlane((unsigned)rem * rs_bytes);
        r[3] = 0x00020000u;
        return r;
    };
    // cold start only: K(t) -> K slot `ks`, V(t) -> V slot `vs`
    auto dma_k_all = [&](int rec, int t, int ks) {
        const u32x4 r = tile_desc(PF64(rec, F_KB_LO), t, PF(rec, F_LK), k_rs_bytes);
        const unsigned l0 = k_lds_wave + (unsigned)(ks * KSLOT);
        dma_piece_first(l0, r, koff[0]);
        dma_piece(l0 + KPIECE, r, piece_off<4>(koff[0], k_rs_bytes));
        dma_piece(l0 + 2 * KPIECE, r, piece_off<8>(koff[0], k_rs_bytes));
        dma_piece(l0 + 3 * KPIECE, r, piece_off<12>(koff[0], k_rs_bytes));
    };
    auto dma_v_all = [&](int rec, int t, int vs) {
        const u32x4 r = tile_desc(PF64(rec, F_VB_LO), t, PF(rec, F_LK), v_rs_bytes);
        const unsigned l0 = v_lds_wave + (unsigned)(vs * S::kTileBytes);
        dma_piece_first(l0, r, voff[0]);
        dma_piece(l0 + 1024, r, piece_off<16>(voff[0], v_rs_bytes));
        dma_piece(l0 + 2048, r, piece_off<32>(voff[0], v_rs_bytes));
        dma_piece(l0 + 3072, r, piece_off<48>(voff[0], v_rs_bytes));
    };
    // this wave's 64 rows of piece c's Q block -> its staging area (16 pieces of 1 KiB); rows at or beyond Sq fetch nothing
    auto dma_q_stage = [&](int rec) {
        const vattn_attn_params& p = args();
        const unsigned q_rs_bytes = (unsigned)p.q_row_stride * 2u;
        const unsigned qvoff = (unsigned)opaque(lane & 31) * q_rs_bytes + (unsigned)(opaque(lane >> 5) << 4);      // row l31 of a 32-row block, d = 8 g .. (+ 32 kk bytes in the instruction)
        const int qw0 = PF(rec, F_QWG0) + wave * 64;
        const unsigned long long a = (unsigned long long)((const T*)p.q + (p.q_start ? 0 : (int64_t)PF(rec, F_B) * p.q_batch_stride) +
                                                          ((long long)PF64(rec, F_QF_LO) + qw0) * p.q_row_stride + (int64_t)PF(rec, F_H) * p.q_head_stride);
        int rows = PF(rec, F_SQ) - qw0;
        rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
        r[2] = __builtin_amdgcn_readfirstlane(rows > 0 ? (unsigned)(rows - 1) * q_rs_bytes + 256u : 0u);
        r[3] = 0x00020000u;
        asm volatile("s_nop 4" ::: "memory");                  // descriptor SGPRs written by readfirstlane -> VMEM
        // 16 pieces of 1 KiB: lane i's 16 bytes (row l31 of query block qc, d = 16 kk + 8 g ..) land at M0 + 16 i — the layout the fragment
        // registers want, so the read-back is lane-linear.  (The 32 kk bytes ride in the lane offset, not in the instruction's offset
        // field: that field also moves the LDS address.)
        const unsigned v1 = qvoff + 32u * q_rs_bytes;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            dma_piece(q_lds_wave + 1024u * kk, r, qvoff + 32u * kk);
            dma_piece(q_lds_wave + 8192u + 1024u * kk, r, v1 + 32u * kk);
        }
    };

    const float escale = p.softmax_scale * kLog2e;
    V8 qf[2][KK];
    // staging -> the Q^T fragment registers (accumulator half)
    auto q_from_stage = [&]() {
#pragma unroll
        for (int qc = 0; qc < 2; qc++)
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                qf[qc][kk] = *(const V8*)(smem + kQStage + wave * 16384 + (qc * 8 + kk) * 1024 + lane * 16);
                asm volatile("" : "+a"(qf[qc][kk]));
            }
    };
    // cold start: Q straight from global memory (prefill64_kernel's prologue)
    auto q_from_global = [&](int rec) {
        const int c_sq = PF(rec, F_SQ);
        const int l31 = opaque(lane & 31), g = opaque(lane >> 5);
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const int my_q = PF(rec, F_QWG0) + wave * 64 + 32 * qc + l31;
            const T* qptr = (const T*)p.q + (p.q_start ? 0 : (int64_t)PF(rec, F_B) * p.q_batch_stride) + ((long long)PF64(rec, F_QF_LO) + my_q) * p.q_row_stride +
                            (int64_t)PF(rec, F_H) * p.q_head_stride;
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (my_q < c_sq) v = *(const uint4*)(qptr + 16 * kk + 8 * g);
                qf[qc][kk] = as_v8<V8>(v);
                asm volatile("" : "+a"(qf[qc][kk]));
            }
        }
    };

    f32x16 o[DB][2];
    float nmsub[2];
    float l_acc[2][2];
    auto reset_acc = [&]() {
#pragma unroll
        for (int i = 0; i < DB; i++)
#pragma unroll
            for (int qc = 0; qc < 2; qc++) {
                o[i][qc] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // the zeros are born in the accumulator file: O^T is carried around a loop that resets it here, and a zero that enters
                // that cycle as an ordinary vector value makes the whole cycle ordinary — 128 copies into the accumulator file at
                // the loop header of every iteration
                asm volatile("" : "+a"(o[i][qc]));
            }
#pragma unroll
        for (int qc = 0; qc < 2; qc++)
#pragma unroll
            for (int a4 = 0; a4 < 2; a4++) l_acc[qc][a4] = 0.f;
    };

    // LDS fragment addressing (prefill64_kernel's)
    const unsigned kfrag_lane = (unsigned)((l31 >> 2) * KPIECE + (l31 & 3) * 16 + g * 64);
    auto kfrag = [&](const char* ksm, int f) -> V8 {
        const int kk = f >> 1, kb = f & 1;
        return *(const V8*)(ksm + kb * 8 * KPIECE + kk * 128 + kfrag_lane);
    };
    const int i16 = lane & 15, dh = (lane >> 4) & 1;
    const unsigned vfrag_lane = (unsigned)((4 * g + (i16 >> 2)) * 64 + (16 * dh + 4 * (i16 & 3)) * 2);
    auto vfrag = [&](const char* vsm, int f) -> V8 {
        const int ks = f >> 2, db = f & 3;
        const char* a1 = vsm + db * S::kVSubBytes + (16 * ks) * 64 + vfrag_lane;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, a1 + 8 * 64));
        return join_tr<V8>(lo, hi);
    };
    // masks tile tt of the piece whose (visible keys, first row of this wave + bottom-right offset) are (Lk_, qoff_)
    // kend_: one past the last key of the PIECE (64 x its last tile + 64): a piece shorter than three tiles is stepped over three anyway
    // (below), and the tiles behind its own — another share's keys, or nothing — must count for nothing
    auto mask_tile = [&](int tt, f32x16 (&s)[2][2], int Lk_, int qoff_, int kend_) {
        const int n0 = tt * PF_BN;
        const int l31 = opaque(lane & 31), g = opaque(lane >> 5);
        const int cap = min(Lk_, kend_) - 1;
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const int lim = causal ? min(cap, qoff_ + 32 * qc + l31) : cap;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = n0 + 32 * kb + 8 * (r >> 2) + 4 * g + (r & 3);
                    if (key > lim) s[kb][qc][r] = -INFINITY;
                }
        }
    };
    auto row_max = [&](const f32x16 (&s)[2][2], int qc) -> float {
        float m0 = fmaxf(s[0][qc][0], s[1][qc][0]);
#pragma unroll
        for (int r = 1; r < 16; r++) m0 = fmaxf(fmaxf(m0, s[0][qc][r]), s[1][qc][r]);
        return fmaxf(m0, swap_halves(m0));
    };
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
    auto pack_p = [&](const f32x16 (&pt)[2][2], int ks, int qc) -> V8 {
        V8 r;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = X::cvt(pt[ks >> 1][qc][8 * (ks & 1) + j]);
        return r;
    };

    f32x16 sc[2][2];
    f32x16 sd[2][2];

    auto GE = [](int e) { return e < NA ? 1 + (e * 30) / NA : 33 + ((e - NA) * 17) / (32 - NA); };
#define P64_X0(cur, e) cur[(e) >> 4][((e) >> 2) & 1][8 * (((e) >> 3) & 1) + 2 * ((e) & 3)]
#define P64_X1(cur, e) cur[(e) >> 4][((e) >> 2) & 1][8 * (((e) >> 3) & 1) + 2 * ((e) & 3) + 1]
    // (round 6, as in prefill64_kernel: the scale in a REAL scalar register, the V^T fragments of phase B read in phase A's tail, the LDS-DMA
    // pieces' distances in the loads' scalar offset, the row-max chain started from its first link: a VALU instruction beside the MFMAs costs
    // 7.3 cycles of this wave, a scalar one 0.3 — profiles/r06_p64_price_list.txt)
    const unsigned escale_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, escale));
    auto softmax_stages = [&](int Gp, f32x16 (&cur)[2][2]) {
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int qc = (e >> 2) & 1;
            if (GE(e) - 1 == Gp)
                asm("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)) : "s"(escale_s), "v"(nmsub[qc]));
            if (GE(e) == Gp) asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(P64_X0(cur, e)), "+v"(P64_X1(cur, e)));
            if (GE(e) + 1 == Gp)
                asm("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(l_acc[qc][0]), "+v"(l_acc[qc][1]) : "v"(P64_X0(cur, e)), "v"(P64_X1(cur, e)));
        }
    };

    // ---- the DMA stream's scalars (prefill64_kernel's; the descriptors live in s[92:95] / s[96:99]) ----
    const unsigned k_tile_b = (unsigned)PF_BN * k_rs_bytes, v_tile_b = (unsigned)PF_BN * v_rs_bytes;
    int k_rows_left = 0, v_rows_left = 0;
    auto bound = [](int rows, unsigned rs) -> unsigned {
        int r;
        asm("s_min_i32 %0, %1, 64\n\ts_max_i32 %0, %0, 0" : "=s"(r) : "s"(rows) : "scc");
        return (unsigned)r * rs;
    };
    u32x4 rk = {0u, 0u, 0u, 0x00020000u};
    u32x4 rv = {0u, 0u, 0u, 0x00020000u};
    // points the running descriptors at tile t of a piece (cold start: K(tb+2) / V(tb+1); re-base: the next piece's K(tb) / V(tb))
    auto k_rebase = [&](unsigned long long base, int t, int Lk_) {
        const unsigned long long a = base + (unsigned long long)t * k_tile_b;
        k_rows_left = Lk_ - t * PF_BN;
        const unsigned bnd = bound(k_rows_left, k_rs_bytes);
        asm volatile("s_mov_b32 s92, %1\n\ts_and_b32 s93, %2, 0xffff\n\ts_mov_b32 s94, %3\n\ts_mov_b32 s95, 0x00020000"
                     : "=&{s[92:95]}"(rk) : "s"((unsigned)a), "s"((unsigned)(a >> 32)), "s"(bnd) : "scc");
    };
    auto v_rebase = [&](unsigned long long base, int t, int Lk_) {
        const unsigned long long a = base + (unsigned long long)t * v_tile_b;
        v_rows_left = Lk_ - t * PF_BN;
        const unsigned bnd = bound(v_rows_left, v_rs_bytes);
        asm volatile("s_mov_b32 s96, %1\n\ts_and_b32 s97, %2, 0xffff\n\ts_mov_b32 s98, %3\n\ts_mov_b32 s99, 0x00020000"
                     : "=&{s[96:99]}"(rv) : "s"((unsigned)a), "s"((unsigned)(a >> 32)), "s"(bnd) : "scc");
    };
    // one tile before tile t of a piece, so that the step's own advance lands on tile t (the previous fetch through the descriptor was issued
    // a step ago; the descriptor is not read again before that advance)
    auto k_prebase = [&](unsigned long long base, int t, int Lk_) {
        const unsigned long long a = base + (unsigned long long)t * k_tile_b - k_tile_b;
        k_rows_left = Lk_ - t * PF_BN + PF_BN;
        asm volatile("s_mov_b32 s92, %1\n\ts_and_b32 s93, %2, 0xffff\n\ts_mov_b32 s94, 0\n\ts_mov_b32 s95, 0x00020000"
                     : "=&{s[92:95]}"(rk) : "s"((unsigned)a), "s"((unsigned)(a >> 32)) : "scc");
    };
    auto v_prebase = [&](unsigned long long base, int t, int Lk_) {
        const unsigned long long a = base + (unsigned long long)t * v_tile_b - v_tile_b;
        v_rows_left = Lk_ - t * PF_BN + PF_BN;
        asm volatile("s_mov_b32 s96, %1\n\ts_and_b32 s97, %2, 0xffff\n\ts_mov_b32 s98, 0\n\ts_mov_b32 s99, 0x00020000"
                     : "=&{s[96:99]}"(rv) : "s"((unsigned)a), "s"((unsigned)(a >> 32)) : "scc");
    };
    unsigned vs_cur = 0, vs_dma = 2 * S::kTileBytes;

    // ---- the scalars of the steady state: tile t of `cur` (whose scores the coming step turns into probabilities), rem = its steps still
    // to go, the first tile that needs masking, its visible keys, (first query row of this wave + bottom-right offset), one past its last
    // key.  EVERY piece is stepped over at least three tiles — the re-base points of the seam need three steps; the tiles behind a
    // shorter piece's own (even a piece without any: stale list, empty sequence) are masked out whole, P = 0 — so every piece chains
    // into its successor and only the first piece of a queue starts cold.  rem_k / rem_v / rem_s: the values of rem at which a step
    // re-bases the K / V descriptor to the NEXT piece's first tile / is the seam (3, 2, 1, or -1 on the queue's last piece: never). ----
    int cur = load_piece(q_idx), nx = 0;          // lane records (load_piece)
    int t = 0, rem = 0, t_mask = 0, Lk_c = 0, qoff_c = 0, kend_c = 0;
    int rem_k = -1, rem_v = -1, rem_s = -1;
    float bx0 = 0.f, bx1 = 0.f;                   // row maxima of the tile a seam scored (after masking)
    bool seam_now = false;                        // the step in flight scores the NEXT piece's first tile

    // The step is prefill64_kernel's but for three things: the piece's counters move inside a gap, the row maxima of the tile scored are
    // left where a seam can take them, and a seam's max-growth test is void.  The re-base of the K / V descriptors costs the step
    // nothing: the glue in front of a re-base step sets the descriptor to ONE TILE BEFORE the next piece's first tile, and the step's
    // ordinary one-tile move (groups 21 / 23) lands on it.
    auto step = [&](const int par, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], V8& kf0, V8& kf1, V8& kf2) {
        const int s_cur = par;
        const char* ksm = smem + (s_cur ^ 1) * KSLOT;
        const char* ksm_next = smem + s_cur * KSLOT;
        const char* vsm = smem + VBASE + vs_cur;
        const unsigned lk0 = k_lds_wave + (unsigned)((s_cur ^ 1) * KSLOT);
        unsigned lv0 = 0;
        V8 pf[2][2];
        V8 kf[RING];
        V8 vf[RING];
        SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int f = i >> 1, qc = i & 1;
            if (f < RING - 1) {
                const V8 a = f == 0 ? kf0 : (f == 1 ? kf1 : kf2);
                if (i < 4) M::qk_first_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
                else M::qk_acc_a(nxt[f & 1][qc], a, qf[qc][f >> 1]);
            } else if (i < 4) M::qk_first(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            else M::qk_acc(nxt[f & 1][qc], kf[f % RING], qf[qc][f >> 1]);
            if ((i & 1) == 0 && f + RING - 1 < 2 * KK) kf[(f + RING - 1) % RING] = kfrag(ksm, f + RING - 1);
            softmax_stages(i, cur);
            if (i == 13) pf[0][0] = pack_p(cur, 0, 0);
            if (i == 15) pf[0][1] = pack_p(cur, 0, 1);
            if (i == 17) lv0 = v_lds_wave + vs_dma;
            if (i == 19) asm volatile("s_mov_b32 %1, %0\n\ts_add_u32 %0, %0, %2\n\ts_cmp_eq_u32 %0, %3\n\ts_cselect_b32 %0, 0, %0"
                                      : "+s"(vs_cur), "=&s"(vs_dma) : "i"(S::kTileBytes), "i"(3 * S::kTileBytes) : "scc");
            // (at a re-base point the descriptor was set back by one tile before the step: the same move lands on the next piece's first tile)
            if (i == 21) k_rsrc_advance(rk, k_rows_left, k_tile_b, k_rs_bytes);
            if (i == 23) v_rsrc_advance(rv, v_rows_left, v_tile_b, v_rs_bytes);
            // the piece's counters move on inside a gap (pinned: scalar C++ would be sunk behind the step's last MFMA, where nothing
            // hides it): t = the tile being scored, rem = tiles left after this step
            if (i == 25) asm volatile("s_add_u32 %0, %0, 1\n\ts_sub_u32 %1, %1, 1" : "+s"(t), "+s"(rem) : : "scc");
            // V(t) landed a step ago: its first fragments are asked for while the last S' MFMAs run (the K ring has stopped reading at i = 24)
            if (i == 26) vf[0] = vfrag(vsm, 0);
            if (i == 28) vf[1] = vfrag(vsm, 1);
            if (i == 30 && RING > 3) vf[2] = vfrag(vsm, 2);
            SCHED_FENCE();
        }
        float mx0 = -INFINITY, mx1 = -INFINITY, g0 = -INFINITY, g1 = -INFINITY, grow = -INFINITY;
        SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int f = j >> 1, ks = j >> 3, qc = j & 1;
            if (j == BJ) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            M::pv(o[f & 3][qc], vf[f % RING], pf[ks & 1][qc]);
            if ((j & 1) == 0 && f + RING - 1 < 16) vf[(f + RING - 1) % RING] = vfrag(vsm, f + RING - 1);
            softmax_stages(32 + j, cur);
            if (ks < 3 && (j & 7) == 4) pf[(ks + 1) & 1][0] = pack_p(cur, ks + 1, 0);
            if (ks < 3 && (j & 7) == 6) pf[(ks + 1) & 1][1] = pack_p(cur, ks + 1, 1);
            if (j >= MS && j < MS + 16) {
                const int r = j - MS;
                if (r == 0) {
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
                asm("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6" : "=&v"(g0), "=&v"(g1) : "v"(mx0), "v"(mx1), "s"(escale_s), "v"(nmsub[0]), "v"(nmsub[1]));
            }
            if (j == MS + 19) asm("v_max_f32 %0, %1, %2" : "=v"(grow) : "v"(g0), "v"(g1));
            if (j == dma_gap(0)) dma_piece_at<0>(lk0, rk, koff[0]);
            if (j == dma_gap(1)) dma_piece_so<KPIECE, 4>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(2)) dma_piece_so<2 * KPIECE, 8>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(3)) dma_piece_so<3 * KPIECE, 12>(lk0, rk, koff[0], k_rs_bytes);
            if (j == dma_gap(4)) dma_piece_at<0>(lv0, rv, voff[0]);
            if (j == dma_gap(5)) dma_piece_so<1024, 16>(lv0, rv, voff[0], v_rs_bytes);
            if (j == dma_gap(6)) dma_piece_so<2048, 32>(lv0, rv, voff[0], v_rs_bytes);
            if (j == dma_gap(7)) dma_piece_so<3072, 48>(lv0, rv, voff[0], v_rs_bytes);
            if (j == 27) kf0 = kfrag(ksm_next, 0);
            if (j == 28) kf1 = kfrag(ksm_next, 1);
            if (j == 29 && RING > 3) kf2 = kfrag(ksm_next, 2);
            SCHED_FENCE();
        }
        // the tile just scored may need masking (at the seam the mask scalars are already the next piece's: VATTN_GLUE)
        if (__builtin_expect(t >= t_mask, 0)) {
            mask_tile(t, nxt, Lk_c, qoff_c, kend_c);
            mx0 = row_max(nxt, 0);
            mx1 = row_max(nxt, 1);
            g0 = __builtin_fmaf(mx0, escale, nmsub[0]);
            g1 = __builtin_fmaf(mx1, escale, nmsub[1]);
            grow = fmaxf(g0, g1);
        }
        if (__builtin_expect(seam_now, 0)) {      // the row maxima of the next piece's first tile (after masking): its reference
            bx0 = mx0;
            bx1 = mx1;
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(grow > kDeferLog2) != 0, 0) && !seam_now) {      // (at the seam `grow` compares two pieces: void)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            SCHED_FENCE();
            raise_max(0, fmaxf(g0, 0.f));
            raise_max(1, fmaxf(g1, 0.f));
            SCHED_FENCE();
            asm volatile("s_nop 3" ::: "memory");
        }
    };

    // ---- epilogue of a finished piece (prefill64_kernel's): O^T[d = 32*db + 8*(r>>2) + 4*g + (r&3)][query] ----
    auto epilogue = [&](int rec) {
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last PV results readable
        SCHED_FENCE();
        const vattn_attn_params& p = args();
        const int l31 = opaque(lane & 31), g = opaque(lane >> 5);
        struct { int b, h, q_wg0, Sq, it_row; long long q_first; } c = {PF(rec, F_B), PF(rec, F_H), PF(rec, F_QWG0), PF(rec, F_SQ), PF(rec, F_ROW), (long long)PF64(rec, F_QF_LO)};
        const bool partial = c.it_row >= 0;
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const int my_q = c.q_wg0 + wave * 64 + 32 * qc + l31;
            const float l_loc = l_acc[qc][0] + l_acc[qc][1];
            const float l_tot = l_loc + swap_halves(l_loc);
            const float inv = (l_tot == 0.f || l_tot != l_tot) ? 1.f : 1.f / l_tot;
            const float m_log2 = -nmsub[qc];
            if (my_q < c.Sq && partial) {
                const int64_t row = (int64_t)c.it_row + (my_q - c.q_wg0);
                float* opart = (float*)p.workspace + row * HD;
                float* lpart = (float*)p.workspace + (int64_t)p.pf_part_rows * HD;
#pragma unroll
                for (int db = 0; db < DB; db++) {
#pragma unroll
                    for (int tq = 0; tq < 4; tq++) {
                        f32x4 w;
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = o[db][qc][4 * tq + e] * inv;
                        *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                    }
                    SCHED_FENCE();      // one 16-register accumulator block at a time: the loop around this code has no registers to lend
                }
                if (g == 0) lpart[row] = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_log2 + __log2f(l_tot));
            } else if (my_q < c.Sq) {
                T* optr = (T*)p.out + (p.q_start ? 0 : (int64_t)c.b * p.o_batch_stride) + (c.q_first + my_q) * p.o_row_stride + (int64_t)c.h * p.o_head_stride;
                {      // 16-byte stores (the launch keeps prefill64_kernel for output strides that are not multiples of 8 elements)
#pragma unroll
                    for (int db = 0; db < DB; db++) {
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
                        SCHED_FENCE();
                    }
                }
                if (p.softmax_lse && g == 0) {
                    const float lse = (l_tot == 0.f) ? INFINITY : (m_log2 + __log2f(l_tot)) * 0.6931471805599453f;
                    p.softmax_lse[((int64_t)c.b * p.h + c.h) * p.seqlen_q + my_q] = lse;
                }
            }
        }
        SCHED_FENCE();
    };

    // ---- kernel start: the V ring holds FINITE data from here on (a key row past a sequence's end has probability exactly 0, and
    // 0 x NaN would poison O; later pieces find the previous pieces' tiles there — finite) ----
    {
        if (DYN && wave == 0) {      // the first ticket, synchronously (nothing else is in flight yet), in front of the barrier below
            draw();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish();
        }
        const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < (3 * S::kTileBytes) / (256 * 16); i++) *(uint4*)(smem + VBASE + (i * 256 + tid) * 16) = z;
        __syncthreads();
    }

    V8 kfa, kfb, kfc;
    bool have_next = false;

    // the mask scalars of piece `rec`, whose first tile is t
    auto piece_scalars = [&](int rec) {
        const int nt_true = PF(rec, F_NT);
        t_mask = min(t_mask_of(rec), nt_true);                  // (tiles behind the piece's own are masked whole)
        Lk_c = PF(rec, F_LK);
        qoff_c = PF(rec, F_QWG0) + wave * 64 + PF(rec, F_OFF);
        kend_c = nt_true * PF_BN;
    };
    // start of a piece: its step count; resolve the next piece of the queue and start its Q block on its way to the staging area
    auto begin_piece = [&]() {
        rem = max(PF(cur, F_NT) - t, 3);
        if (DYN) next_idx = __builtin_amdgcn_readfirstlane(*lds_word);      // published two barriers ago (kernel start / the glue at rem == 2)
        else next_idx = q_idx + 1 < q_end ? q_idx + 1 : -1;
        have_next = next_idx >= 0;
        rem_k = rem_v = rem_s = -1;
        if (have_next) {
            nx = load_piece(next_idx);
            rem_k = 3;
            rem_v = 2;
            rem_s = 1;
            dma_q_stage(nx);
            if (DYN && wave == 0) draw();      // the ticket after next: arrives under the coming step, published at rem == 2
        }
    };
    // The glue in front of a step.  Steady state: ONE compare.  The last three steps of a piece (rem <= 3) and the step after its last
    // (rem <= 0): first the finished piece's epilogue — then its successor is current: the stream goes on, S'(its first tile) sits in
    // the score buffer the last step wrote, its next tiles are in flight, t and the mask scalars are its own since the seam —; then the
    // re-base points (descriptor one tile before the next piece's first: the step's own advance lands on it); at the seam the next
    // piece's Q block (whose DMA the previous step's vmcnt(0) waited for) moves into the fragment registers, and the tile the step
    // scores is the next piece's first: its mask scalars from here on (t is incremented inside the step).
#define VATTN_GLUE()                                                                                                                \
    if (__builtin_expect(rem <= 3, 0)) {                                                                                            \
        if (rem <= 0) {                                                                                                             \
            epilogue(cur);                                                                                                          \
            if (!have_next) break;                                                                                                  \
            q_idx = next_idx;                                                                                                       \
            cur = nx;                                                                                                               \
            reset_acc();                                                                                                            \
            nmsub[0] = (bx0 == -INFINITY) ? 0.f : -bx0 * escale;      /* softmax.h: a fully masked row keeps a zero reference */    \
            nmsub[1] = (bx1 == -INFINITY) ? 0.f : -bx1 * escale;                                                                    \
            seam_now = false;                                                                                                       \
            begin_piece();                                                                                                          \
        }                                                                                                                           \
        if (rem == rem_k) k_prebase(PF64(nx, F_KB_LO), PF(nx, F_TB), PF(nx, F_LK));                                                 \
        if (rem == rem_v) {                                                                                                         \
            v_prebase(PF64(nx, F_VB_LO), PF(nx, F_TB), PF(nx, F_LK));                                                               \
            if (DYN && wave == 0) publish();      /* (drawn at the start of this piece; the step in between waited vmcnt(0)) */      \
        }                                                                                                                           \
        if (rem == rem_s) {                                                                                                         \
            seam_now = true;                                                                                                        \
            q_from_stage();                                                                                                         \
            t = PF(nx, F_TB) - 1;                                                                                                   \
            piece_scalars(nx);                                                                                                      \
        }                                                                                                                           \
    }

    // ---- cold start of the queue's first piece (prefill64_kernel's prologue) ----
    reset_acc();
    t = PF(cur, F_TB);
    piece_scalars(cur);
    dma_k_all(cur, t, 0);
    dma_v_all(cur, t, 0);
    dma_k_all(cur, t + 1, 1);
    q_from_global(cur);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // this wave's pieces of K(tb) landed; V(tb), K(tb+1) may still fly
    __builtin_amdgcn_s_barrier();
    {
        const char* ksm = smem;
#pragma unroll
        for (int f = 0; f < 2 * KK; f++) {
            const V8 a = kfrag(ksm, f);
#pragma unroll
            for (int qc = 0; qc < 2; qc++) {
                if (f < 2) M::qk_first(sc[f & 1][qc], a, qf[qc][f >> 1]);
                else M::qk_acc(sc[f & 1][qc], a, qf[qc][f >> 1]);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        SCHED_FENCE();
        if (t >= t_mask) mask_tile(t, sc, Lk_c, qoff_c, kend_c);
#pragma unroll
        for (int qc = 0; qc < 2; qc++) {
            const float mx = row_max(sc, qc);
            nmsub[qc] = (mx == -INFINITY) ? 0.f : -mx * escale;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // also: every wave is done with K(tb) (the prologue's S')
    dma_k_all(cur, t + 2, 0);
    dma_v_all(cur, t + 1, 1);
    k_rebase(PF64(cur, F_KB_LO), t + 2, Lk_c);                        // the running descriptors at step entry: K(t+2), V(t+1)
    v_rebase(PF64(cur, F_VB_LO), t + 1, Lk_c);
    kfa = kfrag(smem + KSLOT, 0);
    kfb = kfrag(smem + KSLOT, 1);
    kfc = kfrag(smem + KSLOT, 2);
    begin_piece();

    for (;;) {
        VATTN_GLUE()
        step(0, sc, sd, kfa, kfb, kfc);      // even stream position: scores in sc, S' of the following tile into sd
        VATTN_GLUE()
        step(1, sd, sc, kfa, kfb, kfc);      // odd stream position
    }
#undef VATTN_GLUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // trailing DMA retired: nothing may land in the LDS of a later workgroup
#undef P64_X0
#undef P64_X1
#undef PF
#undef PF64
}

template <typename T, bool DYN> static void launch64p_t(const vattn_attn_params* p, hipStream_t st, int* ctr) {
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)prefill64p_kernel<T, DYN, 24, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem64p);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((prefill64p_kernel<T, DYN, 24, 4>), dim3((unsigned)p->pf_num_wg), dim3(256), kSmem64p, st, *p, ctr);
}

// pf_wg_first given: the host-assigned queues; else the drawn ones (ctr: queue_counters(st), attn_api.hip)
void launch_prefill64p(const vattn_attn_params* p, hipStream_t st, int* ctr) {
    const bool bf = p->dtype == VATTN_DTYPE_BF16;
    if (p->pf_wg_first) {
        if (bf) launch64p_t<__bf16, false>(p, st, nullptr);
        else launch64p_t<_Float16, false>(p, st, nullptr);
    } else {
        if (bf) launch64p_t<__bf16, true>(p, st, ctr);
        else launch64p_t<_Float16, true>(p, st, ctr);
    }
}

}  // namespace vattn_k
