// prefill_body.h — the 8/4-wave x 32-row prefill workgroup (see prefill_kernels.hip for the overview) as a device function plus
// the kernel that maps a grid onto it.  Included by prefill_kernels.hip and hybrid_kernels.hip.
#pragma once
#include "attn_common.h"

namespace vattn_k {

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

// WAVES waves per workgroup, each owning QC blocks of 32 query rows (BM = 32*QC*WAVES rows per workgroup).
// QC = 2 halves the LDS fragment traffic per flop (each K / V^T fragment read feeds two MFMAs) at the price
// of a 512-register budget (one wave per SIMD).
// MSUM: the softmax denominator is accumulated by the matrix pipe (one extra MFMA per 16 keys with an all-ones A
// fragment, no LDS read) instead of 32 dependent v_add per tile: the kernel is VALU/issue-bound, the matrix pipe has slack.
// The work of ONE workgroup — query block qb of head h of batch entry b, key-range share `split` of nsplit — as a device function:
// prefill_kernel below maps blockIdx to it; hybrid_kernel (hybrid_kernels.hip) calls it from a persistent loop.
template <typename T, int HD, bool USE_TR, int WAVES, int QC, bool MSUM>
__device__ __forceinline__ void prefill_body(const vattn_attn_params& p, const int b, const int h, const int qb, const int split, const int nsplit, char* smem,
                                             int* merge_counter = nullptr, int* s_ticket = nullptr, const int merge_mode = 0) {
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
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    const int hk = h / (p.h / p.h_k);                          // GQA: head h uses kv head h / (Hq/Hkv)
    // loaded values are wave-uniform; readfirstlane makes that provable (descriptors must live in SGPRs)
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    const int Lk = min(p.seqlen_k, __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew));   // never beyond the cache view
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
        if (p.rotary_cos_sin && my_q < Sq) {
            // fused RoPE: query row i sits at position (visible keys - Sq) + i; slot (g, j) of k-step kk is element 16*kk + 8*g + j,
            // so an element and its partner d + HD/2 live in the same lane (k-steps kk and kk + KK/2)
#pragma unroll
            for (int kk = 0; kk < KK / 2; kk++) {
                V8 c, s;
                rope_load<T>(p, (int64_t)(off + my_q), 16 * kk + 8 * g, c, s);
                rope8<T>(qf[qc][kk], qf[qc][kk + KK / 2], c, s);
            }
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
                    if (kLab && merge_mode == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) store_dev(opart + 32 * db + 8 * tq + 4 * g + e, w[e]);
                    } else {
                        *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                    }
                }
            if (g == 0) {
                const float lv = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_run[qc] * sc + __log2f(l_tot));
                if (kLab && merge_mode == 2) store_dev(lpart + row, lv);
                else lpart[row] = lv;
            }
        } else if (my_q < Sq) {
            T* optr = (T*)p.out + (p.q_start ? 0 : (int64_t)b * p.o_batch_stride) + (q_first + my_q) * p.o_row_stride + (int64_t)h * p.o_head_stride;
            if ((((p.o_row_stride | p.o_head_stride | p.o_batch_stride) & 7) == 0) && !ABL(7)) {
                // 16-byte stores: lane l (g = 0) and lane l + 32 (g = 1) hold d..d+3 and d+4..d+7 of the SAME row for every
                // 8-wide d group tq; one v_permlane32_swap per dword hands the g = 0 lane the whole even group and the g = 1
                // lane the whole odd group -> 8 x 16 B per lane instead of 16 x 8 B (the store tail is issue-bound)
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
                const float lse = (l_tot == 0.f) ? INFINITY : (m_run[qc] * p.softmax_scale + __logf(l_tot));
                p.softmax_lse[((int64_t)b * p.h + h) * p.seqlen_q + my_q] = lse;
            }
        }
    }
    // single-launch merge of the key-range shares (attn_common.h): the workgroup that completes the block's last share merges them
    if (kLab && nsplit > 1 && merge_counter != nullptr)
        prefill_release_and_merge<T, HD>(p, nsplit, b, h, q_wg0, min(Sq, q_wg0 + BM), q_first, merge_counter, s_ticket, merge_mode);
}

template <typename T, int HD, bool USE_TR, int WAVES, int QC, bool MSUM>
__global__ __launch_bounds__(64 * WAVES, (QC == 2 || HD > 128) ? 1 : 2) void prefill_kernel(vattn_attn_params p, int order, int nqb, int nsplit, int* done, int merge_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_ticket;
    int b, h, qb, split;
    if (!wg_to_work(p, order, nqb, nsplit, b, h, qb, split)) return;
    // done: one zeroed counter per (sequence, head, query block) = single-launch merge of the key-range shares; NULL = combine_rows_kernel
    prefill_body<T, HD, USE_TR, WAVES, QC, MSUM>(p, b, h, qb, split, nsplit, smem,
                                                 done ? done + ((int64_t)b * p.h + h) * nqb + qb : nullptr, &s_ticket, merge_mode);
}

}  // namespace vattn_k
