// Fused prefill || decode for hybrid batches — the CDNA4 analogue of the reference's POD-Attention kernel
// (/root/reference/pod_attn/pod_attn/fused_fwd_kernel.h: one CUDA launch whose CTAs are typed prefill / decode by an SM-aware
// counter, so that both operation types are resident on every SM; call site
// /root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_pod_wrapper.py:121-203).
//
// ONE launch of 2 x (number of CUs) PERSISTENT 256-thread workgroups, two per CU (256 registers per lane, 64 KiB of LDS each).
// A workgroup reads its CU's identity (HW_ID / XCC_ID hardware registers) and takes a ticket from that CU's arrival counter:
// the first workgroup to land on a CU prefers PREFILL work, the second prefers DECODE work — every CU then holds one matrix-bound
// wave per SIMD beside one HBM-bound wave per SIMD, which is the co-location two HIP streams cannot promise (a kernel's
// workgroups fill whole CUs first).  Work is handed out by two device-side queues (atomic counters): prefill items are the
// 128-row query blocks (heaviest first), decode items the (sequence, kv head, head block, KV split) tuples; a workgroup whose
// preferred queue has run dry serves the other one, so the launch ends when both are empty, whatever the mix.
// The bodies are the product kernels' own device functions (prefill_body.h: 4 waves x 32 rows; decode_body.h), so the arithmetic
// — and therefore parity — is that of the stand-alone launches.  The split-KV merge of the decode part happens in the same launch:
// the workgroup that completes the last split of a (sequence, kv head, head block) merges its partials (release/acquire through
// __threadfence and the group's counter).
// Control words live at the head of the caller's workspace; they must be zero before the FIRST launch and the kernel leaves
// them zero (the last workgroup to leave resets them), so back-to-back launches on one stream need no memset.
// ROUND 4 — CLOSED: the fused launch is LAB-ONLY (-DVATTN_LAB, tools/lab/libvattn_lab.so).  Three rounds of measurements (fused 0.26-0.75x,
// CU-masked streams 0.42-0.93x, two plain streams 0.88-1.19x of the serial order) say that on MI355X each of the two
// stand-alone kernels already owns what bounds it — HBM for decode, board power for prefill — so co-residency creates no capacity; and the
// last candidate, time-slicing the two phases at workgroup granularity inside one persistent launch, can only remove one launch boundary
// (~2 us) and the decode phase's ramp per layer from launches of 0.5-2 ms: < 1 %, against the >= 5 % the row was asked to show.  The
// PRODUCT library therefore implements the C entry point (the reference's POD call site binds it) as what measures best: the plan-chosen
// prefill launch, then the device-planned decode launch, back to back on the caller's stream.
#include "decode_body_lab.h"
#include "prefill_body_lab.h"

namespace vattn_k {

#ifndef VATTN_LAB
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t hybrid_workspace_bytes(const vattn_attn_params* pp, const vattn_attn_params* pd) {
    return al256(prefill_workspace_bytes(pp)) + al256(decode_workspace_bytes(pd)) + 256;
}
int launch_hybrid(const vattn_attn_params* pp, const vattn_attn_params* pd, void* ws, hipStream_t st) {
    if (pp->dtype != pd->dtype) return fail(VATTN_K_ERR_INVALID, "prefill and decode parts must have the same dtype");
    if (pp->seqlen_q < 2 || pd->seqlen_q != 1) return fail(VATTN_K_ERR_INVALID, "hybrid launch: first part must be a prefill (seqlen_q > 1), second a decode (seqlen_q == 1)");
    if (pp->k_new) return fail(VATTN_K_ERR_UNSUPPORTED, "hybrid launch: append the prefill chunk's keys/values with cache_flat first");
    if (!ws) return fail(VATTN_K_ERR_INVALID, "hybrid launch needs its workspace (vattn_hybrid_workspace_bytes)");
    vattn_attn_params a = *pp, b = *pd;
    a.workspace = ws;
    b.workspace = (char*)ws + al256(prefill_workspace_bytes(pp));
    int rc = launch_prefill_form(&a, st);
    if (rc) return rc;
    return launch_decode_form(&b, st);
}
#else

constexpr int HY_CU_SLOTS = 2048;                    // (xcc, se, sh, cu) keys
constexpr int HY_CTL_INTS = 4 + HY_CU_SLOTS;         // next[2], exited, pad, arrivals[HY_CU_SLOTS]; then done[HY_DONE_CAP]
// The merge counters done[(sequence, kv head, head block)] live in a region of FIXED capacity between the control words and the
// split partials: the partials' offset must not depend on the decode batch (a workspace is reused by later launches with other
// batch sizes, and a counter that lands on bytes an earlier launch filled with fp32 partials never reaches num_splits - 1: the
// merge would silently not run).  Every counter is reset by the workgroup that merges its group, so the region stays zero.
constexpr int HY_DONE_CAP = 1 << 16;

template <typename T>
__global__ __launch_bounds__(256, 2) void hybrid_kernel(vattn_attn_params pp, vattn_attn_params pd, int* ctl, int n_pre, int n_dec, int nqb,
                                                        int dsplits, int gblocks, int fused_append, int role_mode, int merge_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_it[2];
    const int tid = threadIdx.x;
    int* done = ctl + HY_CTL_INTS;
    if (tid == 0) {
        // HW_ID bits 8..15 = cu_id, sh_id, se_id; XCC_ID bits 0..3 = the XCD (gfx940+ hardware register 20)
        const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const unsigned key = (((xcc & 7u) << 8) | (hw & 255u)) & (HY_CU_SLOTS - 1);
        const int ticket = atomicAdd(&ctl[4 + key], 1);
        // role_mode 0: by arrival order on the CU (product); 1 / 2: every workgroup prefers prefill / decode (A/B measurements)
        s_it[1] = role_mode == 1 ? 0 : role_mode == 2 ? 1 : (ticket & 1);
    }
    __syncthreads();
    const int pref = s_it[1];
    for (;;) {
        __syncthreads();                              // everyone has read s_it and is done with the previous item's LDS
        if (tid == 0) {
            int r = pref;
            int it = atomicAdd(&ctl[r], 1);
            if (it >= (r ? n_dec : n_pre)) {          // my queue is dry: help the other one
                r ^= 1;
                it = atomicAdd(&ctl[r], 1);
                if (it >= (r ? n_dec : n_pre)) it = -1;
            }
            s_it[0] = it;
            s_it[1] = r;
        }
        __syncthreads();
        const int it = s_it[0], r = s_it[1];
        if (it < 0) break;
        if (r == 0) {
            // heaviest query blocks first; heads of one kv group are neighbours in the queue (their K/V prefix is shared through L2)
            const int h = it % pp.h;
            const int t = it / pp.h;
            prefill_body<T, 128, true, 4, 1, false>(pp, t % pp.b, h, nqb - 1 - t / pp.b, 0, 1, smem);
        } else {
            const int split = it % dsplits;
            int t = it / dsplits;
            const int gb = t % gblocks;
            t /= gblocks;
            const int hk = t % pd.h_k, b = t / pd.h_k;
            decode_body<T, 128, true, 1>(pd, dsplits, gblocks, fused_append, split, hk, gb, b, smem, merge_mode);
            // release the partial, take the group's ticket, the last one merges (decode_body.h: one agent-scope fence per WORKGROUP —
            // a fence per wave costs 4x that on this multi-XCD part)
            if (dsplits > 1) decode_release_and_merge<T, 128, 1>(pd, dsplits, hk, gb, b, &done[(b * pd.h_k + hk) * gblocks + gb], &s_it[0], merge_mode);
        }
    }
    // leave the control words zero for the next launch
    __syncthreads();
    if (tid == 0) s_it[0] = atomicAdd(&ctl[2], 1);
    __syncthreads();
    if (s_it[0] == (int)gridDim.x - 1) {
        for (int i = tid; i < HY_CTL_INTS; i += 256) ctl[i] = 0;
    }
}

static int hybrid_decode_splits(const vattn_attn_params* pd, int gblocks) {
    if (pd->num_splits > 0) return pd->num_splits > 48 ? 48 : pd->num_splits;
    const long groups = (long)pd->b * pd->h_k * gblocks;
    const int tiles = (pd->seqlen_k + pd->seqlen_knew + DC_BN - 1) / DC_BN;
    long cap = tiles / 4;                            // at least one 32-key tile per wave and split
    if (cap < 1) cap = 1;
    if (cap > 48) cap = 48;
    long want = (4 * 512 + groups - 1) / groups;     // ~4 queue items per resident workgroup: the tail stays short
    if (want > cap) want = cap;
    return want < 1 ? 1 : (int)want;
}

static size_t hybrid_ctl_bytes() {
    const size_t ints = (size_t)HY_CTL_INTS + (size_t)HY_DONE_CAP;
    return ((ints * sizeof(int)) + 255) & ~(size_t)255;
}

size_t hybrid_workspace_bytes(const vattn_attn_params* pp, const vattn_attn_params* pd) {
    (void)pp;
    const int gblocks = (pd->h / pd->h_k + 15) / 16;
    const int ds = hybrid_decode_splits(pd, gblocks);
    return hybrid_ctl_bytes() + (ds > 1 ? (size_t)ds * pd->b * pd->h * (pd->d + 1) * sizeof(float) : 0);
}

template <typename T> static int launch_hybrid_t(const vattn_attn_params* pp, const vattn_attn_params* pd, void* ws, hipStream_t st) {
    const int gblocks = (pd->h / pd->h_k + 15) / 16;
    const int ds = hybrid_decode_splits(pd, gblocks);
    vattn_attn_params d2 = *pd;
    d2.workspace = (char*)ws + hybrid_ctl_bytes();
    if ((long)pd->b * pd->h_k * gblocks > HY_DONE_CAP) return fail(VATTN_K_ERR_UNSUPPORTED, "hybrid launch: more than 65536 (sequence, kv head, head block) decode groups");
    const int nqb = (pp->seqlen_q + 127) / 128;
    const long n_pre = (long)nqb * pp->b * pp->h, n_dec = (long)pd->b * pd->h_k * gblocks * ds;
    if (n_pre > 0x7fffffffL / 2 || n_dec > 0x7fffffffL / 2) return fail(VATTN_K_ERR_INVALID, "hybrid batch too large for the 32-bit work queues");
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const size_t smem = PfSmem<128>::kTotal;          // >= the decode body's 33 KiB
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)hybrid_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, PfSmem<128>::kTotal);
        return true;
    }();
    (void)once;
    const int fused_append = (pd->k_new && pd->seqlen_knew == 1) ? 1 : 0;
    const int role_mode = (pp->variant >> 12) & 3;
    hipLaunchKernelGGL((hybrid_kernel<T>), dim3(2 * cus), dim3(256), smem, st, *pp, d2, (int*)ws, (int)n_pre, (int)n_dec, nqb, ds, gblocks,
                       fused_append, role_mode, (pd->variant & 1024) ? 2 : 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    return VATTN_K_OK;
}

int launch_hybrid(const vattn_attn_params* pp, const vattn_attn_params* pd, void* ws, hipStream_t st) {
    if (pp->d != 128 || pd->d != 128) return fail(VATTN_K_ERR_UNSUPPORTED, "the fused prefill||decode launch is built for head dimension 128");
    if (pp->dtype != pd->dtype) return fail(VATTN_K_ERR_INVALID, "prefill and decode parts must have the same dtype");
    if (pp->seqlen_q < 2 || pd->seqlen_q != 1) return fail(VATTN_K_ERR_INVALID, "hybrid launch: first part must be a prefill (seqlen_q > 1), second a decode (seqlen_q == 1)");
    if (pp->k_new) return fail(VATTN_K_ERR_UNSUPPORTED, "hybrid launch: append the prefill chunk's keys/values with cache_flat first");
    if (pd->k_new && pd->seqlen_knew != 1) return fail(VATTN_K_ERR_UNSUPPORTED, "hybrid launch: the decode part appends exactly one row per sequence");
    if (!ws) return fail(VATTN_K_ERR_INVALID, "hybrid launch needs its workspace (vattn_hybrid_workspace_bytes, zero-filled once)");
    return pp->dtype == VATTN_DTYPE_F16 ? launch_hybrid_t<_Float16>(pp, pd, ws, st) : launch_hybrid_t<__bf16>(pp, pd, ws, st);
}
#endif  // VATTN_LAB

}  // namespace vattn_k
