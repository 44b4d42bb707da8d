// Decode form (seqlen_q == 1) of flash_attn_with_kvcache on gfx950: GQA group packed into the MFMA N dimension, split-KV over
// the context, K fragments loaded straight from HBM into MFMA operand registers, V through a wave-private LDS transpose
// stage, fp32 online softmax, in-workgroup merge of the 4 waves, new K/V row appended in-kernel, LSE-weighted combine across
// splits (combine_kernel; flash_fwd_kernel.h:1116-1297).  Call-site semantics: flash_api.cpp:1367-1378,1451-1454,1558-1560.
#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// decode (seqlen_q == 1): split-KV
// ============================================================================================

constexpr int DC_WAVES = 4;
constexpr int DC_BN = 32;     // keys per wave tile

// workspace layout: float o_accum[splits][b][h][d]; float lse_accum[splits][b][h]  (log2 domain, scaled)
template <typename T, int HD, bool USE_TR>
__global__ __launch_bounds__(64 * DC_WAVES, HD > 128 ? 2 : 3) void decode_kernel(vattn_attn_params p, int num_splits, int gblocks, int fused_append) {
    using X = Tr<T>;
    using V8 = typename X::v8;
    constexpr int KK = HD / 32;          // k-steps of S^T (16x16x32)
    constexpr int DB = HD / 16;          // 16-wide d blocks of O^T
    constexpr int CPR = HD / 8;          // 16-byte chunks per row
    constexpr int VPASS = (DC_BN * CPR) / 64;
    constexpr int V_WAVE_BYTES = DC_BN * HD * 2;        // [d/16][32 keys][16 d] sub-tiles, 32-byte rows
    constexpr int VSUB = DC_BN * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int g4 = lane >> 4;

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
        const float* src = oacc + row * HD + tid;
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < num_splits; s++) acc += wsm[s] * src[(int64_t)s * sstride * HD];
        ((T*)p.out)[(int64_t)b * p.o_batch_stride + (int64_t)q * p.o_row_stride + (int64_t)hh * p.o_head_stride + tid] = Tr<T>::cvt(acc * inv);
    }
    if (p.softmax_lse && tid == 0)
        p.softmax_lse[((int64_t)b * p.h + hh) * sq + q] = (wsum == 0.f) ? INFINITY : (mxs + __log2f(wsum)) * 0.6931471805599453f;
}


// Split count for the decode form.  The kernel is built for 3 workgroups per CU (<= 168 VGPRs, 33 KiB LDS), i.e.
// 768 resident workgroups on 256 CUs; like the reference's heuristic (flash_api.cpp:258-323) pick the smallest
// split count whose last "round" of workgroups is nearly full, but against THIS chip's residency.
int pick_splits(const vattn_attn_params* p, int gblocks) {
    if (p->num_splits > 0) return p->num_splits > 128 ? 128 : p->num_splits;
    const long wg = (long)p->b * p->h_k * gblocks;
    const long slots = 768;
    const int max_len = p->seqlen_k + p->seqlen_knew;
    const int tiles = (max_len + DC_BN - 1) / DC_BN;
    long cap = tiles / 4;                       // at least one 32-key tile per wave and split
    if (cap < 1) cap = 1;
    // a split shorter than ~700 keys costs more in prologue / merge / combine than it returns: B1@32k 20.4 us at 32-48 splits
    // vs 26 us at 128; short contexts still want one tile per wave (B1@2k: 16 splits 11 us vs 24 us unsplit)
    if (cap > 48) cap = 48;
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

template <typename T, int HD> int launch_decode_t(const vattn_attn_params* p, hipStream_t st) {
    const bool use_tr = (p->variant & 1) == 0;
    const int G = p->h / p->h_k;
    const int gblocks = (G + 15) / 16;
    const int splits = pick_splits(p, gblocks);
    if (splits > 1 && !p->workspace) return fail(VATTN_K_ERR_INVALID, "split-KV decode needs a workspace");
    dim3 grid(splits, p->h_k * gblocks, p->b), block(64 * DC_WAVES);
    if (gblocks > 1 && !(p->variant & 64)) {           // sibling head blocks share an XCD (variant bit 6: plain 3-D grid, for A/B)
        const long w = (long)splits * p->h_k * p->b;
        grid = dim3((unsigned)(((w + 7) / 8) * 8 * gblocks));
    }
    const size_t smem = (size_t)DC_WAVES * 16 * HD * 4 + DC_WAVES * 16 * 4 * 2;   // merge area >= V staging (4*8 KiB)
    const int fused_append = (p->k_new && p->seqlen_knew == 1) ? 1 : 0;
    if (p->k_new && !fused_append) launch_append(p, st);        // seqlen_knew > 1: separate append launch
    if (use_tr)
        hipLaunchKernelGGL((decode_kernel<T, HD, true>), grid, block, smem, st, *p, splits, gblocks, fused_append);
    else
        hipLaunchKernelGGL((decode_kernel<T, HD, false>), grid, block, smem, st, *p, splits, gblocks, fused_append);
    if (splits > 1) hipLaunchKernelGGL((combine_kernel<T, HD>), dim3(p->b * p->h), dim3(128), 0, st, *p, splits, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

int launch_decode_form(const vattn_attn_params* p, hipStream_t st) {
    const bool f16 = p->dtype == VATTN_DTYPE_F16;
    if (p->d == 64) return f16 ? launch_decode_t<_Float16, 64>(p, st) : launch_decode_t<__bf16, 64>(p, st);
    return f16 ? launch_decode_t<_Float16, 128>(p, st) : launch_decode_t<__bf16, 128>(p, st);
}

size_t decode_workspace_bytes(const vattn_attn_params* p) {
    const int G = p->h / p->h_k;
    const int gblocks = (G + 15) / 16;
    const int splits = pick_splits(p, gblocks);
    if (splits <= 1) return 0;
    return (size_t)splits * p->b * p->h * (p->d + 1) * sizeof(float);
}

}  // namespace vattn_k
