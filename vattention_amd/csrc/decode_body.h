// decode_body.h — the split-KV decode workgroup (see decode_kernels.hip for the overview) as a device function plus the kernel
// that maps a grid onto it.  Included by decode_kernels.hip and hybrid_kernels.hip.
#pragma once
#include "attn_common.h"

namespace vattn_k {

constexpr int DC_WAVES = 4;
constexpr int DC_BN = 32;     // keys per wave tile

// workspace layout: float o_accum[splits][b][h][d]; float lse_accum[splits][b][h]  (log2 domain, scaled)
// The work of ONE workgroup — split `split` of kv head hk, head block gb, sequence b — as a device function: decode_kernel below
// maps blockIdx to it; hybrid_kernel (hybrid_kernels.hip) calls it from a persistent loop.
template <typename T, int HD, bool USE_TR>
__device__ __forceinline__ void decode_body(const vattn_attn_params& p, const int num_splits, const int gblocks, const int fused_append,
                                            const int split, const int hk, const int gb, const int b, char* smem) {
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
    const int slot = __builtin_amdgcn_readfirstlane(p.cache_batch_idx ? p.cache_batch_idx[b] : b);
    int Lk = __builtin_amdgcn_readfirstlane((p.cache_seqlens ? p.cache_seqlens[b] : p.seqlen_k) + p.seqlen_knew);
    // never beyond the rows of the cache VIEW: the append skips such rows, and the rows behind them may be another slot's or sit on
    // unmapped virtual pages (the wrapper asserts cache_len + new <= rows on the host, where it knows the lengths)
    Lk = Lk > p.seqlen_k ? p.seqlen_k : Lk;

    // each sequence divides ITS OWN length evenly over the splits (balanced for ragged batches)
    const int ntiles_total = (Lk + DC_BN - 1) / DC_BN;
    const int tiles_per_split = (ntiles_total + num_splits - 1) / num_splits;
    const int tile_begin = split * tiles_per_split;
    const int tile_end = min(ntiles_total, tile_begin + tiles_per_split);

    // Fused append (seqlen_knew == 1): the new K/V row sits at key index Lk-1.  Every workgroup that reads the tile
    // holding it substitutes the row from k_new/v_new in registers; the gb == 0 workgroup also stores it into the
    // cache (flash_attn_interface.py:1168-1176: append, then attend).  No inter-workgroup ordering is needed.
    const int new_key = fused_append ? Lk - 1 : -1;
    const int new_tile = fused_append ? new_key / DC_BN : -1;

    const int row_head = gb * 16 + l15;                 // query head within the group handled by this lane's column
    const bool row_valid = row_head < G;
    const int h = hk * G + row_head;
    const T* qptr = (const T*)p.q + (int64_t)b * p.q_batch_stride + (int64_t)h * p.q_head_stride;
    const T* kbase = (const T*)p.k_cache + (int64_t)slot * p.k_batch_stride + (int64_t)hk * p.k_head_stride;
    const T* vbase = (const T*)p.v_cache + (int64_t)slot * p.v_batch_stride + (int64_t)hk * p.v_head_stride;

    // Q^T fragments (B operand, n = query head): slot (g4, j) <-> d = 32*kk + 8*g4 + j
    V8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row_valid) v = *(const uint4*)(qptr + 32 * kk + 8 * g4);
        qf[kk] = as_v8<V8>(v);
    }
    // fused RoPE (include/vattn_kernels.h): the query token sits at position Lk - 1; slot (g4, j) of k-step kk is element
    // d = 32*kk + 8*g4 + j, so element d and its partner d + HD/2 live in the SAME lane (k-steps kk and kk + KK/2)
    const bool rope = p.rotary_cos_sin != nullptr;
    if (rope) {
#pragma unroll
        for (int kk = 0; kk < KK / 2; kk++) {
            V8 c, s;
            rope_load<T>(p, (int64_t)(Lk - 1), 32 * kk + 8 * g4, c, s);
            rope8<T>(qf[kk], qf[kk + KK / 2], c, s);
        }
    }

    f32x4 o[DB];
#pragma unroll
    for (int i = 0; i < DB; i++) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.softmax_scale * kLog2e;
    char* vsm = smem + wave * V_WAVE_BYTES;

    uint4 kreg[2][KK], vreg[VPASS];
    const unsigned k_rs_bytes = (unsigned)p.k_row_stride * 2u, v_rs_bytes = (unsigned)p.v_row_stride * 2u;
    const T* kbase_u = uniform_ptr(kbase);
    const T* vbase_u = uniform_ptr(vbase);
    unsigned koff[2], voff[VPASS];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) koff[kb] = (unsigned)(16 * kb + l15) * k_rs_bytes + (unsigned)g4 * 16u;
#pragma unroll
    for (int ps = 0; ps < VPASS; ps++) {
        const int idx = ps * 64 + lane;
        voff[ps] = (unsigned)(idx / CPR) * v_rs_bytes + (unsigned)(idx % CPR) * 16u;
    }
    auto load_tile = [&](int tile) {
        const int k0 = tile * DC_BN;
        int rem = Lk - k0;
        rem = rem < 0 ? 0 : (rem > DC_BN ? DC_BN : rem);
        const __amdgpu_buffer_rsrc_t kr = make_rsrc(kbase_u + (int64_t)k0 * p.k_row_stride, (unsigned)rem * k_rs_bytes);
        const __amdgpu_buffer_rsrc_t vr = make_rsrc(vbase_u + (int64_t)k0 * p.v_row_stride, (unsigned)rem * v_rs_bytes);
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int kk = 0; kk < KK; kk++) kreg[kb][kk] = buf_load16(kr, koff[kb] + 64u * kk);
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) vreg[ps] = buf_load16(vr, voff[ps]);
        if (tile == new_tile) {      // wave-uniform, at most once per workgroup
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
                        kreg[kb][kk] = v;
                        if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(kc + 32 * kk + 8 * g4) = v;
                    }
                }
#pragma unroll
            for (int ps = 0; ps < VPASS; ps++) {
                const int idx = ps * 64 + lane;
                if (k0 + idx / CPR == new_key) {
                    const uint4 v = *(const uint4*)(vn + (idx % CPR) * 8);
                    vreg[ps] = v;
                    if (gb == 0 && new_key < p.seqlen_k) *(uint4*)(vc + (idx % CPR) * 8) = v;
                }
            }
        }
    };

    int tile = __builtin_amdgcn_readfirstlane(tile_begin + wave);
    load_tile(tile < tile_end ? tile : ntiles_total);     // past the end: every lane out of range, no access
    for (; tile < tile_end; tile += DC_WAVES) {
        const int k0 = tile * DC_BN;
        // ---- V: registers -> wave-private LDS ([d/16][key][16 d]) ----
#pragma unroll
        for (int ps = 0; ps < VPASS; ps++) {
            const int idx = ps * 64 + lane;
            const int row = idx / CPR, c = idx % CPR;
            *(uint4*)(vsm + (c >> 1) * VSUB + row * 32 + ((c & 1) << 4)) = vreg[ps];
        }
        // ---- S^T = K.Q^T on the register-resident K fragments ----
        f32x4 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; kk++) s[kb] = X::mfma16(as_v8<V8>(kreg[kb][kk]), qf[kk], s[kb]);
        }
        // prefetch the wave's next tile while this one is being consumed (out of range past the split's end)
        load_tile(tile + DC_WAVES < tile_end ? tile + DC_WAVES : ntiles_total);

        // s[kb][r] = S^T[key = k0 + 16*kb + 4*g4 + r][head row l15]
        if (k0 + DC_BN > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (k0 + 16 * kb + 4 * g4 + r >= Lk) s[kb][r] = -INFINITY;
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) mloc = fmaxf(mloc, s[kb][r]);
        mloc = fmaxf(mloc, xor_shuffle(mloc, 16));
        mloc = fmaxf(mloc, xor_shuffle(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float msub = (m_new == -INFINITY) ? 0.f : m_new * sc;
        const float alpha = fast_exp2(m_run * sc - msub);
        m_run = m_new;
        float psum = 0.f;
        V8 pf;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = fast_exp2(__builtin_fmaf(s[kb][r], sc, -msub));
                psum += e;
                pf[4 * kb + r] = X::cvt(e);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[i][r] *= alpha;

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- O^T += V^T.P^T : A slot (g4, j) <-> key k0 + (j<4 ? 4*g4 + j : 16 + 4*g4 + j-4) ----
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
            o[db] = X::mfma16(a, pf, o[db]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the 4 waves (each holds a partial softmax over its own tiles) ----
    l_run += xor_shuffle(l_run, 16);
    l_run += xor_shuffle(l_run, 32);
    __syncthreads();                                    // all waves are done with their V staging area
    // o[db][r] = O^T[d = 16*db + 4*g4 + r][head row l15]
    float* osm = (float*)smem;                          // [wave][16 rows][HD]
    float* msm = (float*)(smem + DC_WAVES * 16 * HD * 4);   // [wave][16] m, then [wave][16] l
    float* lsm = msm + DC_WAVES * 16;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
        for (int r = 0; r < 4; r++) osm[(wave * 16 + l15) * HD + 16 * db + 4 * g4 + r] = o[db][r];
    if (g4 == 0) {
        msm[wave * 16 + l15] = m_run;
        lsm[wave * 16 + l15] = l_run;
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * HD; idx += 64 * DC_WAVES) {
        const int row = idx / HD, d = idx % HD;
        const int rh = gb * 16 + row;
        if (rh >= G) continue;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < DC_WAVES; w++) mx = fmaxf(mx, msm[w * 16 + row]);
        float acc = 0.f, lsum = 0.f;
        const float mxs = (mx == -INFINITY) ? 0.f : mx * sc;
#pragma unroll
        for (int w = 0; w < DC_WAVES; w++) {
            const float f = fast_exp2(msm[w * 16 + row] * sc - mxs);
            acc += f * osm[(w * 16 + row) * HD + d];
            lsum += f * lsm[w * 16 + row];
        }
        const int hh = hk * G + rh;
        const float inv = (lsum == 0.f || lsum != lsum) ? 1.f : 1.f / lsum;
        if (num_splits == 1) {
            ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)hh * p.o_head_stride + d] = X::cvt(acc * inv);
            if (p.softmax_lse && d == 0)
                p.softmax_lse[(int64_t)b * p.h + hh] = (lsum == 0.f) ? INFINITY : (mx * p.softmax_scale + __logf(lsum));
        } else {
            float* oacc = (float*)p.workspace;
            float* lacc = oacc + (int64_t)num_splits * p.b * p.h * HD;
            const int64_t row_idx = ((int64_t)split * p.b + b) * p.h + hh;
            oacc[row_idx * HD + d] = acc * inv;
            if (d == 0) lacc[row_idx] = (lsum == 0.f) ? -INFINITY : (mxs + __log2f(lsum));   // log2 domain
        }
    }
}

template <typename T, int HD, bool USE_TR>
__global__ __launch_bounds__(64 * DC_WAVES, HD > 128 ? 2 : 3) void decode_kernel(vattn_attn_params p, int num_splits, int gblocks, int fused_append) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int split, hk, gb, b;
    if (gridDim.y == 1 && gridDim.z == 1 && gblocks > 1) {
        // G > 16 query heads per kv head (MQA models): the ceil(G/16) head blocks of one (split, kv head, sequence) read the
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
    decode_body<T, HD, USE_TR>(p, num_splits, gblocks, fused_append, split, hk, gb, b, smem);
}

}  // namespace vattn_k
