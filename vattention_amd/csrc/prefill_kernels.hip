// Prefill form (seqlen_q > 1) of flash_attn_with_kvcache on gfx950: chunked causal attention over virtually contiguous K/V.
//   prefill_kernel      : 8 (or 4) waves x 32 query rows per workgroup, 64-key tiles double-buffered in LDS (register-staged
//                         global loads two tiles ahead), S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_{f16,bf16},
//                         softmax entirely in registers (a lane owns one query column), K tile XOR-swizzled for conflict-free
//                         ds_read_b128, V tile as [d-block][key][32 d] sub-tiles read through ds_read_b64_tr_b16;
//                         optional KV split (fp32 partials merged by combine_rows_kernel) and batched variable-length chunks
//   prefill_ilv_kernel  : the same data flow software-pipelined with a hand-written issue order (variant 12)
// Semantics: /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, flash_api.cpp:1291-1578, mask.h:164-196
// (bottom-right causal), softmax.h:69-157 (fp32 max/sum, exp2, P rounded to the I/O dtype before PV).
#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// prefill
// ============================================================================================

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
                    *(f32x4*)(opart + 32 * db + 8 * tq + 4 * g) = w;
                }
            if (g == 0) lpart[row] = (l_tot == 0.f || l_tot != l_tot) ? -INFINITY : (m_run[qc] * sc + __log2f(l_tot));
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
    const int Lk = min(p.seqlen_k, __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew));   // never beyond the cache view
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
    if (pl.tiling == 6 && (p->q_lens || p->rotary_cos_sin)) pl.tiling = 1;      // the interleaved kernel has no batched-chunk / fused-RoPE form
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
    // d = 128: prefill64_kernel (4 waves x 64 rows, one 256-row workgroup per CU, LDS-DMA ring, in-wave software pipeline; +15-20 % on
    // every shape that gives its workgroups enough key tiles to amortise the longer prologue) whenever its grid — after the same
    // KV split as the 8-wave tiling — fills at least 3/4 of the CUs and every workgroup gets >= 24 key tiles.  Measured
    // (tools/kbench.py, profiles/r02_kbench.txt): Yi-6B 32 k prompt 968 -> 1165 TFLOP/s, 16 k chunk @ 112 k 1012 -> 1190,
    // Llama-70B/TP8 2 k chunk @ 30 k 918 -> 1052; short whole prompts (2 k tokens: 16 tiles per workgroup) and grids that stay
    // under 192 workgroups keep the tilings below.
    if (p->d == 128) {
        const int ns7 = wg8 >= 256 ? 1 : pick(wg8, 256);
        if (wg8 * ns7 >= 192 && tiles / ns7 >= 24) {
            pl.tiling = 7;
            pl.nsplit = ns7;
            return pl;
        }
    }
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

template <typename T, int HD> int launch_prefill_t(const vattn_attn_params* p, hipStream_t st) {
    const bool use_tr = (p->variant & 1) == 0;
    if (p->k_new && p->seqlen_knew > 0) launch_append(p, st);
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
        } else if (pl.tiling == 2) {
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
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

int launch_prefill_form(const vattn_attn_params* p, hipStream_t st) {
    const bool f16 = p->dtype == VATTN_DTYPE_F16;
    if (p->d == 64) return f16 ? launch_prefill_t<_Float16, 64>(p, st) : launch_prefill_t<__bf16, 64>(p, st);
    return f16 ? launch_prefill_t<_Float16, 128>(p, st) : launch_prefill_t<__bf16, 128>(p, st);
}

size_t prefill_workspace_bytes(const vattn_attn_params* p) {
    const int ns = plan_prefill(p).nsplit;
    return ns > 1 ? (size_t)ns * p->b * p->seqlen_q * p->h * (p->d + 1) * sizeof(float) : 0;
}

}  // namespace vattn_k
