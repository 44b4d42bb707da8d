// decode_body.h — the split-KV decode workgroup (see decode_kernels.hip for the overview) as a device function plus the kernels
// that map a grid onto it.  Included by decode_kernels.hip.  PRODUCT source: the measurement scaffolding of rounds 2-5 (in-launch merge
// protocols, XCD-consecutive ranges, per-workgroup clock stamps, fair-share issue priority, weighted heads, alternative shapes' selectors)
// lives in the lab copy, tools/lab/csrc/decode_body_lab.h, which only tools/lab/libvattn_lab.so compiles.
#pragma once
#include <type_traits>

#include "attn_common.h"

namespace vattn_k {

constexpr int DC_WAVES = 4;
constexpr int DC_BN = 32;     // keys per wave tile

// workspace layout: float o_accum[splits][b][h][d]; float lse_accum[splits][b][h]  (log2 domain, scaled)
// The work of ONE workgroup — split `split` of kv head hk, head-block GROUP gb, sequence b — as a device function: decode_kernel below
// maps blockIdx to it; hybrid_kernel (hybrid_kernels.hip) calls it from a persistent loop.
// NB: 16-head blocks per workgroup (round 2).  With G > 16 query heads per kv head the blocks gb*NB .. gb*NB + NB-1 share ONE pass
// over the K/V rows: the K fragments (registers) and the V^T staging (LDS) of a tile feed NB score / output MFMA chains.  NB = 2
// keeps the kernel inside the 168-register budget of three workgroups per CU; wider groups run as ceil(G/32) such workgroups.
// W:  waves per workgroup (4 = 256 threads, three workgroups per CU; 8 = 512 threads, two per CU: 16 waves per CU instead of 12 and a
//     third fewer partials for the same number of resident wave-tiles).
// PF: K/V register sets per wave = tiles a wave keeps in flight (1: the product shape for chip-filling grids, three workgroups per CU; 2:
//     a second set, requested a whole tile earlier — a lone wave then pulls twice the bytes per round trip, which is what bounds
//     launches with FEWER workgroups than the chip has room for: batch 1-4 decode; two workgroups per CU).
// ROPE: -1 = the fused-RoPE path is compiled in and taken when p.rotary_cos_sin is set; 0 = compiled out (the bf16 build of decode_stream_kernel
// for calls without rotation: its fp32 rotation is 12 registers more than three workgroups per CU leave, see launch_decode_stream); 1 = unconditional
template <typename T, int HD, bool USE_TR, int NB = 1, int W = DC_WAVES, int PF = 1, int ROPE = -1>
__device__ __forceinline__ void decode_body(const vattn_attn_params& p, const int num_splits, const int gblocks, const int fused_append,
                                            const int split, const int hk, const int gb, const int b, char* smem,
                                            const int item = -1, const int item_tb = 0, const int item_te = 0,
                                            const int st_mode = 0, const int st_slot = 0, const int st_lk = 0, const unsigned st_block = 0,
                                            const int tstride = 1) {
    // st_mode != 0: a piece [item_tb, item_te) of the device-planned stream decomposition (decode_stream_kernel below).  Slot and visible
    // length come from the workgroup's plan (LDS) instead of two dependent global loads; st_mode 1 = the piece is the whole sequence: the
    // final rows are written; st_mode 2 = a partial, published as one record block at byte offset st_block of the workspace (16-byte
    // stores; merged by the next launch).
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int KK = HD / 32;          // k-steps of S^T (16x16x32)
    constexpr int DB = HD / 16;          // 16-wide d blocks of O^T
    constexpr int CPR = HD / 8;          // 16-byte chunks per row
    constexpr int VPASS = (DC_BN * CPR) / 64;
    constexpr int V_WAVE_BYTES = DC_BN * HD * 2;        // [d/16][32 keys][16 d] sub-tiles, 32-byte rows
    constexpr int VSUB = DC_BN * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int g4 = lane >> 4;

    const int G = p.h / p.h_k;
    int slot, Lk;
    if (st_mode) {
        slot = st_slot;
        Lk = st_lk;
    } else {
        slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
        Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
        // never beyond the rows of the cache VIEW: the append skips such rows, and the rows behind them may be another slot's or sit on
        // unmapped virtual pages (the wrapper asserts cache_len + new <= rows on the host, where it knows the lengths)
        Lk = Lk > p.seqlen_k ? p.seqlen_k : Lk;
    }

    // each sequence divides ITS OWN length evenly over the splits (balanced for ragged batches)
    // (stream mode: an EMPTY sequence still owns one tile of the plan's tile space — fully masked, so that its rows get written)
    const int ntiles_total = (st_mode && Lk <= 0) ? 1 : (Lk + DC_BN - 1) / DC_BN;
    const int tiles_per_split = (ntiles_total + num_splits - 1) / num_splits;
    // item >= 0: a piece [item_tb, item_te) of a length-balanced plan (vattn_decode_plan) instead of split `split` of num_splits
    // tstride > 1 (STRIPED pieces): this workgroup's tiles are tile_begin, tile_begin + tstride, ... — the pieces of a sequence interleave
    // tile by tile instead of each streaming its own contiguous range, so that the workgroups that run together sweep ONE window of the
    // sequence (tools/decode_skew_probe.py: with contiguous ranges megabytes apart, the kv head whose bytes have address bits [9:8] = 01
    // runs 20 % behind the others and the launch waits for it).  Split mode: piece `split` of num_splits; piece mode: item_tb is the piece index.
    const bool striped = tstride > 1;
    const int tile_begin = item >= 0 ? item_tb : (striped ? split : split * tiles_per_split);
    const int tile_end = min(ntiles_total, item >= 0 ? item_te : (striped ? ntiles_total : tile_begin + tiles_per_split));

    // Fused append (seqlen_knew == 1): the new K/V row sits at key index Lk-1.  Every workgroup that reads the tile
    // holding it substitutes the row from k_new/v_new in registers; the gb == 0 workgroup also stores it into the
    // cache (flash_attn_interface.py:1168-1176: append, then attend).  No inter-workgroup ordering is needed.
    const int new_key = fused_append ? Lk - 1 : -1;

    const T* kbase = (const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride;
    const T* vbase = (const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride;

    // Q^T fragments (B operand, n = query head): slot (g4, j) <-> d = 32*kk + 8*g4 + j
    V8 qf[NB][KK];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const int row_head = (gb * NB + nb) * 16 + l15;     // query head within the group handled by this lane's column
        const T* qptr = (const T*)p.q + (int64_t)b * p.q_batch_stride + (int64_t)(hk * G + row_head) * p.q_head_stride;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row_head < G) v = *(const uint4*)(qptr + 32 * kk + 8 * g4);
            qf[nb][kk] = as_v8<V8>(v);
        }
    }
    // fused RoPE (include/vattn_kernels.h): the query token sits at position Lk - 1; slot (g4, j) of k-step kk is element
    // d = 32*kk + 8*g4 + j, so element d and its partner d + HD/2 live in the SAME lane (k-steps kk and kk + KK/2)
    const bool rope = ROPE < 0 ? p.rotary_cos_sin != nullptr : ROPE == 1;
    if (rope) {
#pragma unroll
        for (int kk = 0; kk < KK / 2; kk++) {
            V8 c, s;
            rope_load<T>(p, (int64_t)(Lk - 1), 32 * kk + 8 * g4, c, s);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) rope8<T>(qf[nb][kk], qf[nb][kk + KK / 2], c, s);
        }
    }

    // W > 4 (two 8-wave or one 16-wave workgroup per CU: 128 registers per lane): the Q^T fragments live in LDS (one copy per workgroup,
    // every wave holds the same values) and are re-read for each tile's score MFMAs instead of occupying 4 x KK registers for the
    // whole key walk — LDS bandwidth is idle in this kernel, registers are what bounds the waves (and so the bytes) in flight
    constexpr bool QLDS = W > DC_WAVES;
    char* const qsm = smem + W * 16 * HD * 4 + W * 16 * 4 * 2;
    if constexpr (QLDS) {
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int kk = 0; kk < KK; kk++) *(V8*)(qsm + ((nb * KK + kk) * 64 + lane) * 16) = qf[nb][kk];
        }
        __syncthreads();
    }

    f32x4 o[NB][DB];
    float m_run[NB], l_run[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
        for (int i = 0; i < DB; i++) o[nb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_run[nb] = -INFINITY;
        l_run[nb] = 0.f;
    }
    const float sc = p.softmax_scale * kLog2e;
    char* vsm = smem + wave * V_WAVE_BYTES;

    uint4 kreg[PF][2][KK], vreg[PF][VPASS];
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;
    const T* kbase_u = uniform_ptr(kbase);
    const T* vbase_u = uniform_ptr(vbase);
    // per-lane byte offsets inside a tile: ONE live register each; the row groups of the further instructions are wave-uniform
    // multiples of the row stride added per instruction (in the VGPR offset: the instruction's scalar offset is EXCLUDED from the
    // descriptor's bounds check, and the bound is what keeps the loads off unmapped pages), the k-steps travel in the immediate
    const unsigned koff0 = (unsigned)l15 * k_rs_bytes + (unsigned)g4 * 16u;
    const unsigned voff0 = (unsigned)(lane / CPR) * v_rs_bytes + (unsigned)(lane % CPR) * 16u;
    const unsigned k_kb_step = __builtin_amdgcn_readfirstlane(16u * k_rs_bytes);
    const unsigned v_ps_step = __builtin_amdgcn_readfirstlane((unsigned)(64 / CPR) * v_rs_bytes);
    auto load_tile = [&](const int u, int tile) {      // u: register set, a literal at every call site (unrolled loops)
        const int k0 = tile * DC_BN;
        int rem = Lk - k0;
        rem = rem < 0 ? 0 : (rem > DC_BN ? DC_BN : rem);
        const __amdgpu_buffer_rsrc_t kr = make_rsrc(kbase_u + (int64_t)k0 * p.k_row_stride, (unsigned)rem * k_rs_bytes);
        const __amdgpu_buffer_rsrc_t vr = make_rsrc(vbase_u + (int64_t)k0 * p.v_row_stride, (unsigned)rem * v_rs_bytes);
        // (opaque copies: the per-instruction offsets are re-derived from ONE register each with a v_add per load instead of being
        // hoisted out of the key loop into ten loop-invariant registers — registers are what bounds the waves in flight here)
        unsigned ko = koff0, vo = voff0;
        asm volatile("" : "+v"(ko), "+v"(vo));
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int kk = 0; kk < KK; kk++) kreg[u][kb][kk] = buf_load16(kr, ko + (unsigned)kb * k_kb_step + 64u * kk);
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) vreg[u][ps] = buf_load16(vr, vo + (unsigned)ps * v_ps_step);
    };
    // the new K/V row (fused append) replaces its row of the LAST tile in registers; the gb == 0 workgroup also stores it
    auto substitute_new_row = [&](int k0) {      // (register set 0: the last tile is loaded there)
        const T* kn = (const T*)p.k_new + (int64_t)b * p.knew_batch_stride + (int64_t)hk * p.knew_head_stride;
        const T* vn = (const T*)p.v_new + (int64_t)b * p.vnew_batch_stride + (int64_t)hk * p.vnew_head_stride;
        T* kc = (T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride + (int64_t)new_key * p.k_row_stride;
        T* vc = (T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride + (int64_t)new_key * p.v_row_stride;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
            if (k0 + 16 * kb + l15 == new_key) {
                V8 kn8[KK];
#pragma unroll
                for (int kk = 0; kk < KK; kk++) kn8[kk] = as_v8<V8>(*(const uint4*)(kn + 32 * kk + 8 * g4));
                if (rope) {                  // the new key is rotated before it is attended and before it is stored
#pragma unroll
                    for (int kk = 0; kk < KK / 2; kk++) {
                        V8 c, s;
                        rope_load<T>(p, (int64_t)new_key, 32 * kk + 8 * g4, c, s);
                        rope8<T>(kn8[kk], kn8[kk + KK / 2], c, s);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KK; kk++) {
                    uint4 v;
                    __builtin_memcpy(&v, &kn8[kk], 16);
                    kreg[0][kb][kk] = v;
                    if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(kc + 32 * kk + 8 * g4) = v;
                }
            }
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) {
            const int idx = ps * 64 + lane;
            if (k0 + idx / CPR == new_key) {
                const uint4 v = *(const uint4*)(vn + (idx % CPR) * 8);
                vreg[0][ps] = v;
                if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(vc + (idx % CPR) * 8) = v;
            }
        }
    };
    // One 32-key tile of this wave: V registers -> wave-private LDS, S^T = K.Q^T on the register-resident K fragments, request the
    // wave's next tile into the freed registers, online softmax, O^T += V^T.P^T.  RAGGED: the sequence's last tile (keys at or
    // beyond Lk are masked) — a literal at both call sites, so the steady-state loop carries no mask and no append code.
    auto process_tile = [&](auto ragged_tag, const int u, const int tile, const int next_tile) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int k0 = tile * DC_BN;
        // ---- V: registers -> wave-private LDS ([d/16][key][16 d]) ----
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) {
            const int idx = ps * 64 + lane;
            const int row = idx / CPR, c = idx % CPR;
            *(uint4*)(vsm + (c >> 1) * VSUB + row * 32 + ((c & 1) << 4)) = vreg[u][ps];
        }
        // ---- S^T = K.Q^T (every head block of the group uses the same K fragments) ----
        f32x4 s[NB][2];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            V8 qt[KK];
            if constexpr (QLDS) {
                asm volatile("" ::: "memory");                // a fresh read per tile (not hoisted back into registers)
#pragma unroll
                for (int kk = 0; kk < KK; kk++) qt[kk] = *(const V8*)(qsm + ((nb * KK + kk) * 64 + lane) * 16);
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; kk++) qt[kk] = qf[nb][kk];
            }
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
                s[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KK; kk++) s[nb][kb] = X::mfma16(as_v8<V8>(kreg[u][kb][kk]), qt[kk], s[nb][kb]);
            }
        }
        // request the wave's next tile while this one is being consumed (past the end: every lane out of range, no access)
        if (!RAGGED) load_tile(u, next_tile);

        // s[nb][kb][r] = S^T[key = k0 + 16*kb + 4*g4 + r][head row l15 of block nb]
        V8 pf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            if (RAGGED) {
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (k0 + 16 * kb + 4 * g4 + r >= Lk) s[nb][kb][r] = -INFINITY;
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) mloc = fmaxf(mloc, s[nb][kb][r]);
            mloc = fmaxf(mloc, xor_shuffle(mloc, 16));
            mloc = fmaxf(mloc, xor_shuffle(mloc, 32));
            const float m_new = fmaxf(m_run[nb], mloc);
            const float msub = (m_new == -INFINITY) ? 0.f : m_new * sc;
            const float alpha = fast_exp2(m_run[nb] * sc - msub);
            m_run[nb] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = fast_exp2(__builtin_fmaf(s[nb][kb][r], sc, -msub));
                    psum += e;
                    pf[nb][4 * kb + r] = X::cvt(e);
                }
            l_run[nb] = l_run[nb] * alpha + psum;
#pragma unroll
            for (int i = 0; i < DB; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[nb][i][r] *= alpha;
        }

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- O^T += V^T.P^T : A slot (g4, j) <-> key k0 + (j<4 ? 4*g4 + j : 16 + 4*g4 + j-4); one V^T fragment read per d block ----
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
#pragma unroll
            for (int nb = 0; nb < NB; nb++) o[nb][db] = X::mfma16(a, pf[nb], o[nb][db]);
        }
        __builtin_amdgcn_wave_barrier();
    };

    // This wave's tiles: tile_begin + wave, + W, ... below tile_end.  The sequence's LAST tile (the only one that can be ragged, and
    // the one that holds the appended row) is always the last tile of whichever wave owns it: it is taken out of the steady-state
    // loop and processed after it.
    const int last_tile = ntiles_total - 1;
    const bool special_last = fused_append || (Lk % DC_BN) != 0 || Lk <= 0;
    const int wstep = __builtin_amdgcn_readfirstlane(W * tstride);          // distance between two tiles of one wave
    const int first = __builtin_amdgcn_readfirstlane(tile_begin + wave * tstride);
    const bool own_last = special_last && last_tile >= first && last_tile < tile_end && ((last_tile - first) % wstep) == 0;
    const int loop_end = own_last ? last_tile : tile_end;          // wave-uniform
    if (first < loop_end) {
#pragma unroll
        for (int u = 0; u < PF; u++) load_tile(u, first + u * wstep < loop_end ? first + u * wstep : ntiles_total);
        for (int tile0 = first; tile0 < loop_end; tile0 += PF * wstep) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int tile = tile0 + u * wstep;
                if (tile >= loop_end) break;                 // wave-uniform
                const int nxt = tile + PF * wstep;
                process_tile(std::false_type{}, u, tile, nxt < loop_end ? nxt : ntiles_total);
            }
        }
    }
    if (own_last) {
        load_tile(0, last_tile);
        if (fused_append) substitute_new_row(last_tile * DC_BN);
        process_tile(std::true_type{}, 0, last_tile, ntiles_total);
    }

    // ---- merge the W waves (each holds a partial softmax over its own tiles), one head block after the other ----
    float* osm = (float*)smem;                          // [wave][16 rows][HD]
    float* msm = (float*)(smem + W * 16 * HD * 4);   // [wave][16] m, then [wave][16] l
    float* lsm = msm + W * 16;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        float lr = l_run[nb];
        lr += xor_shuffle(lr, 16);
        lr += xor_shuffle(lr, 32);
        __syncthreads();                                // all waves are done with their V staging area / the previous block's merge
        // o[nb][db][r] = O^T[d = 16*db + 4*g4 + r][head row l15]
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int r = 0; r < 4; r++) osm[(wave * 16 + l15) * HD + 16 * db + 4 * g4 + r] = o[nb][db][r];
        if (g4 == 0) {
            msm[wave * 16 + l15] = m_run[nb];
            lsm[wave * 16 + l15] = lr;
        }
        __syncthreads();
        if (st_mode) {
            // stream decomposition: four columns per thread — 16-byte LDS reads and 16-byte stores (a device-scope store of one dword
            // costs ~6x its share of a 16-byte one: MI355X_MICROARCH, inter-workgroup visibility)
            const __amdgpu_buffer_rsrc_t wsr = make_rsrc(p.workspace, 0x7fffffffu);
            for (int i4 = tid; i4 < 16 * (HD / 4); i4 += 64 * W) {
                const int row = i4 / (HD / 4), d0 = (i4 % (HD / 4)) * 4;
                const int rh = (gb * NB + nb) * 16 + row;
                if (rh >= G) continue;
                float mx = -INFINITY;
#pragma unroll
                for (int w = 0; w < W; w++) mx = fmaxf(mx, msm[w * 16 + row]);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                float lsum = 0.f;
                const float mxs = (mx == -INFINITY) ? 0.f : mx * sc;
#pragma unroll
                for (int w = 0; w < W; w++) {
                    const float f = fast_exp2(msm[w * 16 + row] * sc - mxs);
                    const f32x4 a = *(const f32x4*)&osm[(w * 16 + row) * HD + d0];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] += f * a[e];
                    lsum += f * lsm[w * 16 + row];
                }
                const int hh = hk * G + rh;
                const float inv = (lsum == 0.f || lsum != lsum) ? 1.f : 1.f / lsum;
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] *= inv;
                if (st_mode == 1) {
                    typename X::v4 o4;
#pragma unroll
                    for (int e = 0; e < 4; e++) o4[e] = X::cvt(acc[e]);
                    *(typename X::v4*)((T*)p.out + (int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + d0) = o4;
                    if (p.softmax_lse && d0 == 0)
                        p.softmax_lse[(int64_t)b * p.h + hh] = (lsum == 0.f) ? INFINITY : (mx * p.softmax_scale + __logf(lsum));
                } else {
                    // record block: float o[16 * NB][HD], then float lse[16 * NB] (log2 domain) — see decode_stream_kernel
                    const unsigned r16 = (unsigned)(nb * 16 + row);
                    u32x4 bits;
                    __builtin_memcpy(&bits, &acc, 16);
                    const float lv = (lsum == 0.f) ? -INFINITY : (mxs + __log2f(lsum));
                    __builtin_amdgcn_raw_buffer_store_b128(bits, wsr, (int)(st_block + (r16 * HD + d0) * 4u), 0, 0);
                    if (d0 == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lv), wsr, (int)(st_block + (16u * NB * HD + r16) * 4u), 0, 0);
                }
            }
            continue;
        }
        for (int idx = tid; idx < 16 * HD; idx += 64 * W) {
            const int row = idx / HD, d = idx % HD;
            const int rh = (gb * NB + nb) * 16 + row;
            if (rh >= G) continue;
            float mx = -INFINITY;
#pragma unroll
            for (int w = 0; w < W; w++) mx = fmaxf(mx, msm[w * 16 + row]);
            float acc = 0.f, lsum = 0.f;
            const float mxs = (mx == -INFINITY) ? 0.f : mx * sc;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const float f = fast_exp2(msm[w * 16 + row] * sc - mxs);
                acc += f * osm[(w * 16 + row) * HD + d];
                lsum += f * lsm[w * 16 + row];
            }
            const int hh = hk * G + rh;
            const float inv = (lsum == 0.f || lsum != lsum) ? 1.f : 1.f / lsum;
            if (num_splits == 1 && item < 0) {
                ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + d] = X::cvt(acc * inv);
                if (p.softmax_lse && d == 0)
                    p.softmax_lse[(int64_t)b * p.h + hh] = (lsum == 0.f) ? INFINITY : (mx * p.softmax_scale + __logf(lsum));
            } else {
                float* oacc = (float*)p.workspace;
                float* lacc = oacc + (item >= 0 ? (int64_t)p.num_split_items : (int64_t)num_splits * p.b) * p.h * HD;
                const int64_t row_idx = item >= 0 ? (int64_t)item * p.h + hh : ((int64_t)split * p.b + b) * p.h + hh;
                const float lv = (lsum == 0.f) ? -INFINITY : (mxs + __log2f(lsum));   // log2 domain
                oacc[row_idx * HD + d] = acc * inv;
                if (d == 0) lacc[row_idx] = lv;
            }
        }
    }
}

// ============================================================================================
// Device-planned stream decomposition of a decode batch (round 4)
// ============================================================================================
// The reference's split heuristic (flash_api.cpp:258-323) and rounds 1-3 of this kernel give every sequence of a batch the same number
// of splits; a ragged batch then lasts as long as its longest sequence's share.  Round 3 balanced it with a HOST-built plan — which needs
// the lengths on the host, and the reference's wrapper only has them on the device (vattention_flashattention_wrapper.py:194-205 passes
// `cache_seqlens` / `cache_batch_idx` tensors).  Here the plan is derived ON THE DEVICE, by every workgroup for itself, from those two
// arrays: the 32-key tiles of all sequences of one (kv head, head-block group) form ONE tile space in batch order; workgroup w of the
// nwg workgroups of that group streams tiles [w T, (w + 1) T), T = ceil(total / nwg) — every workgroup reads the same number of bytes
// whatever the lengths are.  A range that crosses sequence boundaries is processed piece by piece (one softmax state per sequence).
//   plan prologue: lane i of every wave loads cache_seqlens[i] and cache_batch_idx[i] (ONE memory latency for both), a wave-wide inclusive
//     scan of the tile counts, a compare + ballot for the first sequence of the range — registers only, no LDS, no barrier, no dependent
//     second load before the K/V stream starts;
//   pieces of sequence b come from the consecutive workgroups first_w .. last_w (closed form from the scan), piece (w, b) publishes
//     its partial as record (w + b) — unique, and consecutive within a sequence;
//   a sequence that lies inside ONE workgroup's range is written directly; the others are merged by decode_stream_combine_kernel (a
//     second launch that re-derives the plan the same way) — no host plan, no upload.  LAB: the workgroup that completes a sequence's
//     last piece merges inside the launch (device-scope ticket per (sequence, kv head)): measured equal or slower, see decode_kernels.hip.
// Partials travel as 16-byte device-scope (write-through) stores — nothing is left dirty in the L2s for the kernel boundary to write
// back — one 128-byte-aligned record block per (record, kv head, group) that exactly one workgroup writes and exactly one reads.
// STRIPED pieces (see decode_body).  Product: the split-KV launch of ONE sequence (profiles/r04_decode_striped_ab.txt: one 128 k sequence
// on 28 / 4 heads 60.5 -> 58.5 us; chip-filling batches +-0, one kv head per GPU -3 %: those keep contiguous pieces).  LAB: variant bit 24
// stripes every uniform decomposition, bit 25 keeps the single sequence contiguous (A/B: tools/decode_striped_ab.py).
__device__ __forceinline__ bool decode_striped(const vattn_attn_params& p) { return p.b == 1; }
constexpr int DC_MAXB = 256;                         // sequences per launch the plan prologue handles (4 per lane of a wave)
// A workgroup whose range crosses into another sequence pays a second prologue / epilogue (partial stores drained, Q fetched, the K/V
// stream restarted): a few microseconds during which its neighbours on the CU keep streaming but IT falls behind — and a one-round
// launch ends with its slowest workgroup.  Every sequence therefore occupies `switch allowance` extra positions of the tile space ahead
// of its first tile: ranges stay equal in POSITIONS, a range that crosses a boundary holds that many fewer real tiles.
constexpr int DC_SWITCH_TILES = 4;
constexpr int DC_MIN_WG_TILES = 8;                   // a workgroup streams at least this many positions (two tiles per wave)
__device__ __forceinline__ int stream_switch_tiles(const vattn_attn_params& p) { return (p.split_reserved & 255) > 0 ? (p.split_reserved & 255) - 1 : DC_SWITCH_TILES; }
__device__ __forceinline__ int stream_tiles_per_wg(int total, int nwg) {
    const int t = (total + nwg - 1) / nwg;
    return t < DC_MIN_WG_TILES ? DC_MIN_WG_TILES : t;
}

// workspace: int2 table[b] = (first record, record count) of every sequence — written by the workgroup that owns the sequence's first
// piece on kv head 0, read by the merge launch (ONE scalar load instead of re-deriving the plan) — then the records, 128-byte aligned
__host__ __device__ __forceinline__ constexpr unsigned stream_table_bytes(int b) { return ((unsigned)b * 8u + 127u) & ~127u; }
template <int NB, int HD> struct StreamRec {
    static constexpr int kFloats = 16 * NB * HD + 32;        // o[16 NB][HD], lse[16 NB] padded to a 128-byte line
};

// What every workgroup of the launch (and of the merge launch) derives from the lengths: the decomposition and, per sequence, which
// records hold its pieces.
//   STREAM (above): workgroup w owns positions [w T, (w + 1) T).
//   UNIFORM: when cutting EVERY sequence into the same number S = nwg / B of pieces of its own length is at least as balanced — the
//   longest piece, ceil(longest sequence / S), is no longer than a stream range — the pieces are taken aligned to the sequences
//   (workgroup w = piece w % S of sequence w / S): equal-length batches (the static trace) then have no workgroup that crosses a
//   sequence boundary, and the decomposition IS the one the reference's heuristic (flash_api.cpp:258-323) would launch.
// Both are pure functions of (lengths, nwg): the decode launch and the merge launch agree without exchanging anything.
struct StreamGeom {
    int T;            // stream: positions per workgroup
    int S;            // uniform: pieces per sequence
    bool uniform;
};
__device__ __forceinline__ StreamGeom stream_geom(const int total, const int maxt, const int B, const int nwg) {
    StreamGeom g;
    g.T = stream_tiles_per_wg(total, nwg);
    g.S = nwg / B;
    g.uniform = g.S >= 1 && (maxt + g.S - 1) / g.S <= g.T;
    if (!g.uniform) {
        // A SHORT tile space is streamed by fewer, longer ranges: a third of the workgroups (one per CU) below 20 positions per
        // workgroup — the rest of the grid exits at once.  The host sized the grid for the longest contexts the cache view allows; how
        // much there really is to read is only known here.  [Measured, 256 ragged sequences on one kv head, mean 1.2 k tokens: 34.8 us
        // with 256 workgroups, 36.2 with 512, 36.4 with 768 (profiles/r04_decode_stream.txt (7)): every range pays a prologue, a record
        // and a share of the merge, and a launch of a few tens of microseconds has no stream to hide them behind.  Between 20 and 64
        // positions two workgroups per CU measured within the noise of three (2.5 k tokens: 59.2 vs 61.7 us; 4.8 k: 122 vs 123-127).]
        const int tpw = total / nwg;
        const int eff = tpw < 20 ? max(1, nwg / 3) : nwg;
        g.T = stream_tiles_per_wg(total, eff);
    }
    return g;
}
// records of sequence b (tiles at positions [excl + X, incl)): first record and count
__device__ __forceinline__ void stream_seq_records(const StreamGeom& g, const int excl, const int incl, const int X, const int b, int& first_rec, int& cnt) {
    if (g.uniform) {
        const int t = incl - excl - X, per = (t + g.S - 1) / g.S;
        first_rec = b * g.S + b;
        cnt = (t + per - 1) / per;
    } else {
        const int first_w = (excl + X) / g.T, last_w = (incl - 1) / g.T;
        first_rec = first_w + b;
        cnt = last_w - first_w + 1;
    }
}

// Plan prologue, per WAVE and in registers (no LDS, no barrier: the four waves of a workgroup derive the same plan side by side and each
// starts its K/V stream as soon as ITS copy is done).  Lane l holds sequences [l << sh, (l + 1) << sh), sh = 0 / 1 / 2 for batches up
// to 64 / 128 / 256: incl = positions of sequences 0..i (inclusive scan of tiles + switch allowance), slot, lk = cache slot and visible
// length.  ONE memory latency (lengths and slots are fetched together), six shuffle steps.
struct StreamPlan {
    int incl[4], slot[4], lk[4];
    int sh, total, maxt;
};
__device__ __forceinline__ void stream_plan_load(const vattn_attn_params& p, const int X, StreamPlan& pl) {
    const int lane = threadIdx.x & 63;
    const int B = p.b;
    pl.sh = B <= 64 ? 0 : B <= 128 ? 1 : 2;
    int sum = 0, mx = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int i = (lane << pl.sh) + e;
        int t = 0;
        pl.lk[e] = 0;
        pl.slot[e] = 0;
        if (e < (1 << pl.sh) && i < B) {
            int lk = (p.cache_seqlens ? p.cache_seqlens[i] : p.seqlen_k);
            lk = (lk < 0 ? 0 : lk) + p.seqlen_knew;
            lk = lk > p.seqlen_k ? p.seqlen_k : lk;                  // never beyond the rows of the cache view (decode_body)
            pl.lk[e] = lk;
            pl.slot[e] = p.cache_batch_idx ? p.cache_batch_idx[i] : i;
            t = (lk > 0 ? (lk + DC_BN - 1) / DC_BN : 1);            // an empty sequence owns one (masked) tile: its rows get written
            mx = max(mx, t);
            t += X;
        }
        sum += t;
        pl.incl[e] = sum;
    }
    int v = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    const int excl = v - sum;
#pragma unroll
    for (int e = 0; e < 4; e++) pl.incl[e] += excl;
    pl.total = __builtin_amdgcn_readlane(v, 63);
    pl.maxt = __builtin_amdgcn_readfirstlane(mx);
}
// element b of a per-lane array (b wave-uniform)
__device__ __forceinline__ int stream_plan_get(const int (&a)[4], const int sh, const int b) {
    const int e = b & ((1 << sh) - 1), l = b >> sh;
    const int v = e == 0 ? a[0] : e == 1 ? a[1] : e == 2 ? a[2] : a[3];
    return __builtin_amdgcn_readlane(v, l);
}
// the sequence whose positions [excl, incl) hold position g (g < total)
__device__ __forceinline__ int stream_plan_find(const StreamPlan& pl, const int g) {
    const int lane = threadIdx.x & 63;
    int lo = __shfl_up(pl.incl[3], 1, 64);
    if (lane == 0) lo = 0;
    int hit = -1;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (lo <= g && g < pl.incl[e]) hit = e;
        lo = pl.incl[e];
    }
    const unsigned long long m = __ballot(hit >= 0);
    const int l = __builtin_ctzll(m);
    return (l << pl.sh) + __builtin_amdgcn_readlane(hit, l);
}

__device__ __forceinline__ void stream_publish_seq(const vattn_attn_params& p, const int b, const int first_rec, const int cnt) {
    int* t = (int*)p.workspace + 2 * b;
    t[0] = first_rec;
    t[1] = cnt;
}

// LSE-weighted merge of records [first_rec, first_rec + cnt) of (kv head hk, group gb) into the output rows of sequence b, by one
// 256-thread workgroup: a thread owns four columns of one head row, walks the records in chunks of CH with every load of a chunk
// in flight at once (device-scope loads: the records were written by other workgroups of this launch) and folds the chunks together
// like an online softmax (running maximum, running weight sum) — any number of records, one pass.
template <typename T, int HD, int NB, int CH = 8, int SCOPE = kDevScope>
__device__ __forceinline__ void decode_stream_merge(const vattn_attn_params& p, const int first_rec, const int cnt, const int hk, const int gb,
                                                    const int gblocks, const int b) {
    using X = Tr<T>;
    constexpr int RF = StreamRec<NB, HD>::kFloats;
    const int tid = threadIdx.x;
    const int G = p.h / p.h_k;
    const __amdgpu_buffer_rsrc_t wsr = make_rsrc(p.workspace, 0x7fffffffu);
    const unsigned blk_stride = (unsigned)p.h_k * (unsigned)gblocks * RF * 4u;                 // bytes between consecutive records
    const unsigned blk0 = stream_table_bytes(p.b) + ((unsigned)first_rec * p.h_k * gblocks + (unsigned)hk * gblocks + gb) * RF * 4u;
    for (int i4 = tid; i4 < 16 * NB * (HD / 4); i4 += 256) {
        const int r16 = i4 / (HD / 4), d0 = (i4 % (HD / 4)) * 4;
        const int rh = gb * NB * 16 + r16;
        if (rh >= G) continue;
        const unsigned o_off = blk0 + (unsigned)(r16 * HD + d0) * 4u, l_off = blk0 + (unsigned)(16 * NB * HD + r16) * 4u;
        float m = -INFINITY, wsum = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < cnt; s0 += CH) {
            float l[CH];
            u32x4 v[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int s = s0 + j < cnt ? s0 + j : cnt - 1;      // (the tail of the last chunk re-reads the last record, weight 0)
                l[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wsr, (int)(l_off + (unsigned)s * blk_stride), 0, SCOPE));
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(wsr, (int)(o_off + (unsigned)s * blk_stride), 0, SCOPE);
            }
            float cm = -INFINITY;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                if (s0 + j >= cnt) l[j] = -INFINITY;
                cm = fmaxf(cm, l[j]);
            }
            const float m_new = fmaxf(m, cm);
            const float mns = (m_new == -INFINITY) ? 0.f : m_new;
            const float r = fast_exp2(m - mns);
            wsum *= r;
#pragma unroll
            for (int e = 0; e < 4; e++) acc[e] *= r;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const float w = fast_exp2(l[j] - mns);
                f32x4 a;
                __builtin_memcpy(&a, &v[j], 16);
                wsum += w;
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] += w * a[e];
            }
            m = m_new;
        }
        const float inv = (wsum == 0.f) ? 0.f : 1.f / wsum;
        const int hh = hk * G + rh;
        typename X::v4 o4;
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = X::cvt(acc[e] * inv);
        *(typename X::v4*)((T*)p.out + (int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + d0) = o4;
        if (p.softmax_lse && d0 == 0)
            p.softmax_lse[(int64_t)b * p.h + hh] = (wsum == 0.f) ? INFINITY : (((m == -INFINITY) ? 0.f : m) + __log2f(wsum)) * 0.6931471805599453f;
    }
}

// nwg: workgroups per (kv head, group) = gridDim.x.  The partials are merged by decode_stream_combine_kernel in a second launch (merging
// inside the launch, XCD-consecutive ranges, per-workgroup clock stamps, fair-share issue priority: tools/lab/csrc/decode_body_lab.h).
template <typename T, int HD, bool USE_TR, int NB, int ROPE = -1>
__global__ __launch_bounds__(64 * DC_WAVES, NB > 1 ? 2 : 3) void decode_stream_kernel(vattn_attn_params p, int gblocks, int fused_append) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_plan[3 * DC_MAXB];                // stream mode: the plan, for the pieces after the first
    const int tid = threadIdx.x;
    // grid (workgroups per head, kv heads): consecutive workgroup ids — consecutive XCDs — stream consecutive ranges of ONE kv head.  [The
    // kv head as the fastest index (the heads of one range dispatched together) measured 5-7 % slower on every shape, profiles/r04_decode_stream.txt.]
    const int hk = blockIdx.y / gblocks, gb = blockIdx.y % gblocks;
    const int nwg = gridDim.x;
    const int w = blockIdx.x;
    const int X = stream_switch_tiles(p);
    constexpr unsigned RB = StreamRec<NB, HD>::kFloats * 4u;
    StreamPlan pl;
    stream_plan_load(p, X, pl);
    const StreamGeom geo = stream_geom(pl.total, pl.maxt, p.b, nwg);
    // the current piece (all wave-uniform): sequence, its slot and visible length, tiles [tb, te), the sequence's records
    int b, slot, lk, tb, te, first_rec, cnt;
    int g0 = 0, g1 = 0;
    // stream mode: the first piece at or after sequence `from` that holds real tiles of this workgroup's range (plan read from LDS)
    auto next_piece = [&](const int from) -> bool {
        for (b = from; b < p.b; b++) {
            const int excl = __builtin_amdgcn_readfirstlane(b ? s_plan[b - 1] : 0);
            if (excl >= g1) return false;
            const int incl = __builtin_amdgcn_readfirstlane(s_plan[b]);
            const int real0 = excl + X;                  // the sequence's tile r sits at position real0 + r
            tb = max(g0, real0) - real0;
            te = min(g1, incl) - real0;
            if (te <= tb) continue;                      // this range only holds (part of) the sequence's switch allowance
            stream_seq_records(geo, excl, incl, X, b, first_rec, cnt);
            slot = __builtin_amdgcn_readfirstlane(s_plan[DC_MAXB + b]);
            lk = __builtin_amdgcn_readfirstlane(s_plan[2 * DC_MAXB + b]);
            return true;
        }
        return false;
    };
    if (geo.uniform) {
        // piece w % S of sequence w / S: each sequence divides ITS OWN tiles into S pieces; everything from the plan's registers
        b = w / geo.S;
        const int sidx = w - b * geo.S;
        if (b >= p.b) return;
        const int excl = b ? stream_plan_get(pl.incl, pl.sh, b - 1) : 0, incl = stream_plan_get(pl.incl, pl.sh, b);
        const int t = incl - excl - X, per = (t + geo.S - 1) / geo.S;
        tb = sidx * per;
        te = min(t, tb + per);
        cnt = (t + per - 1) / per;
        if (te <= tb) return;
        first_rec = b * geo.S + b;
        slot = stream_plan_get(pl.slot, pl.sh, b);
        lk = stream_plan_get(pl.lk, pl.sh, b);
    } else {
        g0 = w * geo.T;
        if (g0 >= pl.total) return;
        g1 = min(pl.total, g0 + geo.T);
        // the range may hold several sequences: the plan moves to LDS (every wave stores the SAME values and reads only after its own
        // stores: no barrier), so that its twelve registers are not carried through the key loops
        const int b0 = stream_plan_find(pl, g0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = ((tid & 63) << pl.sh) + e;
            if (e < (1 << pl.sh) && i < p.b) {
                s_plan[i] = pl.incl[e];
                s_plan[DC_MAXB + i] = pl.slot[e];
                s_plan[2 * DC_MAXB + i] = pl.lk[e];
            }
        }
        if (!next_piece(b0)) return;
    }
    for (;;) {
        const unsigned blk = stream_table_bytes(p.b) + (((unsigned)(w + b) * p.h_k + hk) * gblocks + gb) * RB;
        if (tb == 0 && hk == 0 && gb == 0 && tid == 0) stream_publish_seq(p, b, first_rec, cnt);      // (the owner of the sequence's first piece)
        decode_body<T, HD, USE_TR, NB, DC_WAVES, 1, ROPE>(p, 2, gblocks, fused_append, 0, hk, gb, b, smem, 0, tb, te, cnt == 1 ? 1 : 2, slot, lk, blk);
        if (geo.uniform || !next_piece(b + 1)) return;
        __syncthreads();                                 // the previous piece's in-workgroup merge is done with the LDS
    }
}

// The merge as a second launch: one workgroup per (sequence, kv head x group) merges the sequence's records (nothing to do for sequences
// that one workgroup wrote directly).
template <typename T, int HD, int NB>
__global__ __launch_bounds__(256) void decode_stream_combine_kernel(vattn_attn_params p, int gblocks) {
    const int b = blockIdx.x, hk = blockIdx.y / gblocks, gb = blockIdx.y % gblocks;
    const int* t = (const int*)p.workspace + 2 * b;
    const int first_rec = __builtin_amdgcn_readfirstlane(t[0]), cnt = __builtin_amdgcn_readfirstlane(t[1]);
    if (cnt <= 1) return;
    decode_stream_merge<T, HD, NB, 16, 0>(p, first_rec, cnt, hk, gb, gblocks, b);      // (up to 16 records in ONE round trip)
}

// gblocks = head-block GROUPS per kv head (ceil(ceil(G/16) / NB)).  The partials of a split launch are merged by combine_kernel in a
// second launch (the single-launch merges live in the lab copy).
template <typename T, int HD, bool USE_TR, int NB, int W = DC_WAVES, int PF = 1>
__global__ __launch_bounds__(64 * W, W > 4 ? 4 : (HD > 128 || (HD == 128 && NB > 1) || PF > 1) ? 2 : 3) void decode_kernel(vattn_attn_params p, int num_splits, int gblocks, int fused_append) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int split, hk, gb, b;
    if (p.split_items != nullptr) {
        // length-balanced plan: blockIdx.x = work item (a piece of ONE sequence), blockIdx.y = (kv head, head-block group)
        const vattn_decode_item it = p.split_items[blockIdx.x];
        decode_body<T, HD, USE_TR, NB, W, PF>(p, 2, gblocks, fused_append, it.index_in_seq, blockIdx.y / gblocks, blockIdx.y % gblocks, it.b, smem,
                                          (int)blockIdx.x, it.tile_begin, it.tile_end);
        return;
    }
    if (gridDim.y == 1 && gridDim.z == 1 && gblocks > 1) {
        // G > 32 query heads per kv head (MQA models): the head-block groups of one (split, kv head, sequence) read the
        // SAME K/V rows.  1-D grid laid out so that those sibling workgroups get consecutive slots on ONE XCD (ids 8 apart):
        // the first reader pulls the rows from HBM, the others hit that XCD's L2
        const int L = blockIdx.x;
        const int xcd = L & 7, j = L >> 3;
        gb = j % gblocks;
        const int w = (j / gblocks) * 8 + xcd;             // flattened (split, kv head, sequence)
        const int per_b = num_splits * p.h_k;
        if (w >= per_b * p.b) return;
        b = w / per_b;
        hk = (w % per_b) / num_splits;
        split = w % num_splits;
    } else {
        split = blockIdx.x;
        hk = blockIdx.y / gblocks;
        gb = blockIdx.y % gblocks;
        b = blockIdx.z;
    }
    decode_body<T, HD, USE_TR, NB, W, PF>(p, num_splits, gblocks, fused_append, split, hk, gb, b, smem, -1, 0, 0, 0, 0, 0, 0,
                                          (decode_striped(p) && num_splits > 1) ? num_splits : 1);
}

}  // namespace vattn_k
