// C ABI of the gfx950 attention kernels over virtually-contiguous KV tensors (include/vattn_kernels.h).
//   prefill_kernels.hip : seqlen_q > 1   (chunked causal prefill, KV split, batched variable-length chunks)
//   decode_kernels.hip  : seqlen_q == 1  (split-KV decode with in-kernel append, combine)
//   cache_kernels.hip   : cache_flat / append
// Semantics follow the operator the reference calls (flash_attn_with_kvcache):
//   /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291, flash_api.cpp:1291-1578,
//   mask.h:164-196 (bottom-right causal), softmax.h:69-157 (fp32 max/sum, exp2, P rounded to the
//   I/O dtype before PV), flash_fwd_kernel.h:1116-1297 (split combine).
// Every K/V access is predicated on the sequence's visible length: rows at or beyond it may sit
// on unmapped virtual pages (SURVEY §7 "never touch unmapped VA").
#include <map>
#include <mutex>
#include <utility>
#include "attn_common.h"

namespace vattn_k {

// ============================================================================================
// hardware-layout self test
// ============================================================================================

// Checks, against plain integer arithmetic, the three layout facts the kernels rely on:
//  [0] 32x32x16 C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
//  [1] 16x16x32 C/D map: col = lane&15, row = 4*(lane>>4) + r
//  [2] ds_read_b64_tr_b16: lane i of a 16-lane group receives element (i&3) of the 8-byte chunks
//      addressed by lanes 4*j + (i>>2), j = 0..3, of the same group
//  [3] A/B operands: lane (x = lane&31, g = lane>>5) contributes row/col x with k-slots (g, 0..7) (32x32x16)
//  [4] same for 16x16x32 with g = lane>>4
//  [5] LDS-DMA (`buffer_load_dwordx4 ... lds`, prefill64_kernels.hip): lane i's 16 bytes land at LDS address M0 + 16*i, whatever
//      global offset the lane fetched from (the K-tile swizzle is applied on the global side)
//  [6] (informational, not a failure) what a lane beyond the descriptor's bound does to its 16 LDS bytes:
//      0 = writes zeros, 1 = leaves them untouched, 2 = something else
__global__ void selftest_kernel(int* res, const unsigned* gsrc) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    const int lane = threadIdx.x;
    // [0],[3]: A = one-hot rows, B = one-hot cols with distinct values -> C[m][n] = sum_k A[m][k]B[k][n]
    {
        // A[m][slot] = (m + 1) if slot == (m & 15) else 0 ; B[slot][n] = (n + 1) * 64 + ... keep small ints
        f16x8 a, bq;
        const int x = lane & 31, g = lane >> 5;
        for (int j = 0; j < 8; j++) {
            const int slot = 8 * g + j;                 // logical k index shared by A and B
            a[j] = (_Float16)((slot == (x & 15)) ? (float)(x + 1) : 0.f);      // A[m=x][k]
            bq[j] = (_Float16)((slot == 3) ? 0.f : 0.f);
        }
        // B[k][n=x] = 1 for every k -> C[m][n] = sum_k A[m][k] = m + 1 for every n
        for (int j = 0; j < 8; j++) bq[j] = (_Float16)1.f;
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, c, 0, 0, 0);
        int bad = 0;
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            if (c[r] != (float)(row + 1)) bad = 1;
        }
        // columns: A[m][k] = 1 for all, B[k][n] = (n+1) if k-slot == (n & 15) -> C[m][n] = n + 1
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)1.f;
            bq[j] = (_Float16)(((8 * g + j) == (x & 15)) ? (float)(x + 1) : 0.f);
        }
        f32x16 c2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, c2, 0, 0, 0);
        int bad3 = 0;
        for (int r = 0; r < 16; r++)
            if (c2[r] != (float)(x + 1)) bad3 = 1;
        if (bad) atomicOr(&res[0], 1);
        if (bad3) atomicOr(&res[3], 1);
    }
    {
        f16x8 a, bq;
        const int x = lane & 15, g = lane >> 4;
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)(((8 * g + j) == x) ? (float)(x + 1) : 0.f);
            bq[j] = (_Float16)1.f;
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c, 0, 0, 0);
        int bad = 0;
        for (int r = 0; r < 4; r++)
            if (c[r] != (float)(4 * g + r + 1)) bad = 1;
        for (int j = 0; j < 8; j++) {
            a[j] = (_Float16)1.f;
            bq[j] = (_Float16)(((8 * g + j) == x) ? (float)(x + 1) : 0.f);
        }
        f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c2, 0, 0, 0);
        int bad4 = 0;
        for (int r = 0; r < 4; r++)
            if (c2[r] != (float)(x + 1)) bad4 = 1;
        if (bad) atomicOr(&res[1], 1);
        if (bad4) atomicOr(&res[4], 1);
    }
    {
        // each lane owns the 8-byte chunk at lds[lane*4 .. lane*4+3]; value encodes (lane, element)
        for (int e = 0; e < 4; e++) lds[lane * 4 + e] = (short)(lane * 4 + e);
        __syncthreads();
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, &lds[lane * 4]));
        const int grp = lane >> 4, i = lane & 15;
        int bad = 0;
        for (int j = 0; j < 4; j++) {
            const int src_lane = grp * 16 + 4 * j + (i >> 2);
            if (t[j] != (short)(src_lane * 4 + (i & 3))) bad = 1;
        }
        if (bad) atomicOr(&res[2], 1);
    }
    {
        __shared__ __attribute__((aligned(16))) unsigned dst[2 * 256];      // two 1-KiB pieces
        for (int e = 0; e < 8; e++) dst[lane * 8 + e] = 0xABABABABu;
        __syncthreads();
        // gsrc[w] = w for 512 dwords (2 KiB).  piece 0: lane i fetches 16-byte chunk (i ^ 5); piece 1: descriptor bounded at 512
        // bytes, lane i fetches chunk i (lanes >= 32 are out of range)
        const unsigned long long a = (unsigned long long)gsrc;
        u32x4 r0, r1;
        r0[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r0[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
        r0[2] = 2048u;
        r0[3] = 0x00020000u;
        r1 = r0;
        r1[2] = 512u;
        const unsigned l0 = (unsigned)(size_t)LDS_PTR(unsigned, dst), l1 = l0 + 1024u;
        const unsigned v0 = (unsigned)((lane ^ 5) * 16), v1 = (unsigned)(lane * 16);
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(l0), "s"(r0), "v"(v0) : "memory", "m0");
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(l1), "s"(r1), "v"(v1) : "memory", "m0");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int bad = 0;
        for (int e = 0; e < 4; e++)
            if (dst[lane * 4 + e] != (unsigned)((lane ^ 5) * 4 + e)) bad = 1;
        if (lane < 32)
            for (int e = 0; e < 4; e++)
                if (dst[256 + lane * 4 + e] != (unsigned)(lane * 4 + e)) bad = 1;
        if (bad) atomicOr(&res[5], 1);
        if (lane >= 32) {
            int code = 0;
            for (int e = 0; e < 4; e++) {
                const unsigned x = dst[256 + lane * 4 + e];
                if (x == 0xABABABABu) code |= 1;
                else if (x != 0u) code |= 2;
            }
            if (code) atomicOr(&res[6], code);
        }
    }
}

// [7] (informational) LDS-DMA into an LDS address beyond 64 KiB (the destination base travels in M0): 1 KiB is sent to byte offset
// 72 KiB of a 96-KiB dynamic allocation; reports 1 + the KiB offset the data is found at (73 = where it was sent), 0 = nowhere
__global__ void selftest_dma_high_kernel(int* res, const unsigned* gsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned big[];
    const int lane = threadIdx.x;
    for (int i = lane; i < (96 << 10) / 4; i += 64) big[i] = 0xCDCDCDCDu;
    __syncthreads();
    const unsigned long long a = (unsigned long long)gsrc;
    u32x4 r0;
    r0[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r0[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    r0[2] = 2048u;
    r0[3] = 0x00020000u;
    const unsigned l0 = (unsigned)(size_t)LDS_PTR(unsigned, big) + (72u << 10);
    const unsigned v0 = (unsigned)(lane * 16);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds" : : "s"(l0), "s"(r0), "v"(v0) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        int found = 0;
        for (int kib = 0; kib < 96; kib++) {
            bool ok = true;
            for (int w = 0; w < 256 && ok; w++) ok = big[kib * 256 + w] == (unsigned)w;
            if (ok) { found = kib + 1; break; }
        }
        res[7] = found;
    }
}

// Counters of the single-launch merges (split-KV decode, KV-split prefill) for launches on stream `st`: created (and zeroed, stream-ordered) on first use, never while
// the stream is being captured into a graph (then the caller takes the two-launch form; a warm-up call before capture creates it).
int* merge_counters(hipStream_t st, size_t n_ints) {
    if (!kLab) return nullptr;        // the in-launch merges of the stand-alone launches are measurement scaffolding (measured slower)
    struct Buf { int* p; size_t n; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    Buf& b = bufs[std::make_pair(dev, st)];
    if (b.p && b.n >= n_ints) return b.p;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    const size_t n = n_ints < 16384 ? 16384 : n_ints;
    int* np = nullptr;
    if (hipMalloc((void**)&np, n * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemsetAsync(np, 0, n * sizeof(int), st) != hipSuccess) { (void)hipFree(np); return nullptr; }
    // an older, smaller buffer may still be in use by launches queued on the stream: it is leaked on purpose (a few KiB, at most
    // once per growth step)
    b.p = np;
    b.n = n;
    return np;
}


// Counters of the DRAWN queues of the persistent prefill launch (csrc/prefill64p_kernels.hip): 8 ints per (device, stream), zeroed by the
// caller in front of every drawn launch (launch_prefill_t; the last draw of a queue also resets its counter).  Created on first use; never
// while the stream is being captured into a graph (NULL then: the caller launches one workgroup per piece; a warm-up call before capture
// creates them).
int* queue_counters(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int*> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    auto it = bufs.find(std::make_pair(dev, st));
    if (it != bufs.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    int* np = nullptr;
    if (hipMalloc((void**)&np, 16 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemsetAsync(np, 0, 16 * sizeof(int), st) != hipSuccess) { (void)hipFree(np); return nullptr; }
    bufs[std::make_pair(dev, st)] = np;
    return np;
}

thread_local std::string g_err;
int fail(int code, const char* msg) {
    g_err = msg;
    return code;
}

// the caller's header and ours describe the same block (include/vattn_kernels.h)
bool abi_ok(const vattn_attn_params* p) { return p && p->struct_size == (uint32_t)sizeof(vattn_attn_params) && p->abi_version == VATTN_KERNELS_ABI; }

int validate(const vattn_attn_params* p) {
    if (!p) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    if (!abi_ok(p)) return fail(VATTN_K_ERR_INVALID, "vattn_attn_params: struct_size / abi_version do not match this library (zero the block, then set both from include/vattn_kernels.h)");
    if (!p->q || !p->out || !p->k_cache || !p->v_cache) return fail(VATTN_K_ERR_INVALID, "null tensor pointer");
    if (p->dtype != VATTN_DTYPE_F16 && p->dtype != VATTN_DTYPE_BF16)
        return fail(VATTN_K_ERR_UNSUPPORTED, "FlashAttention only support fp16 and bf16 data type");      // flash_api.cpp:1325-1326
    // d = 256 instantiates but spills (O^T alone is 128 accumulator registers per wave): not shipped until it has its own tiling
    if (p->d != 64 && p->d != 128) return fail(VATTN_K_ERR_UNSUPPORTED, "this build supports head dimensions 64 and 128");
    if (p->b <= 0) return fail(VATTN_K_ERR_INVALID, "batch size must be postive");                       // flash_api.cpp:1353
    if (p->h_k <= 0 || p->h % p->h_k != 0)
        return fail(VATTN_K_ERR_INVALID, "Number of heads in key/value must divide number of heads in query");   // :1355
    if ((p->k_new == nullptr) != (p->v_new == nullptr))
        return fail(VATTN_K_ERR_INVALID, "If key is supplied, value must also be passed in");            // :1452
    if (p->k_new && !p->cache_seqlens)
        return fail(VATTN_K_ERR_INVALID, "If key is supplied, seqlens_k must also be passed in");        // :1453
    if (p->seqlen_q <= 0 || p->seqlen_k < 0) return fail(VATTN_K_ERR_INVALID, "bad sequence lengths");
    if ((p->q_start == nullptr) != (p->q_lens == nullptr)) return fail(VATTN_K_ERR_INVALID, "q_start and q_lens must be given together");
    if (p->q_lens && p->seqlen_q == 1) return fail(VATTN_K_ERR_UNSUPPORTED, "batched chunks need max(q_lens) > 1 (the decode form is already batched)");
    if (p->q_lens && p->k_new) return fail(VATTN_K_ERR_UNSUPPORTED, "batched chunks: append the new keys/values with cache_flat first");
    // 16-byte vector access requirements
    const int64_t strides[] = {p->q_batch_stride, p->q_row_stride, p->q_head_stride, p->k_batch_stride, p->k_row_stride,
                               p->k_head_stride, p->v_batch_stride, p->v_row_stride, p->v_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return fail(VATTN_K_ERR_UNSUPPORTED, "strides must be multiples of 8 elements (16-byte vector access)");
    if (((uintptr_t)p->q | (uintptr_t)p->k_cache | (uintptr_t)p->v_cache | (uintptr_t)p->out) & 15)
        return fail(VATTN_K_ERR_UNSUPPORTED, "tensor base pointers must be 16-byte aligned");
    if (p->rotary_cos_sin) {
        if (p->rotary_dim != p->d) return fail(VATTN_K_ERR_UNSUPPORTED, "fused rotary embedding needs rotary_dim == head dimension");
        if ((p->rotary_row_stride & 7) || ((uintptr_t)p->rotary_cos_sin & 15))
            return fail(VATTN_K_ERR_UNSUPPORTED, "rotary cos/sin rows must be 16-byte aligned");
    }
    if (p->o_row_stride % 4 != 0 || p->o_head_stride % 4 != 0 || p->o_batch_stride % 4 != 0)
        return fail(VATTN_K_ERR_UNSUPPORTED, "output strides must be multiples of 4 elements");
    if (p->pf_items && (p->seqlen_q == 1 || p->d != 128)) return fail(VATTN_K_ERR_INVALID, "pf_items (prefill work list) applies to the prefill form with head dimension 128");
    if (p->pf_num_wg < 0 || (p->pf_wg_first && !p->pf_num_wg) || (p->pf_num_wg && (!p->pf_items || p->pf_num_wg > p->num_pf_items)))
        return fail(VATTN_K_ERR_INVALID, "pf_num_wg (persistent work list) needs pf_items and at most one workgroup per piece; pf_wg_first needs pf_num_wg");
    if (p->split_items && p->seqlen_q != 1) return fail(VATTN_K_ERR_INVALID, "split_items (length-balanced plan) applies to the decode form only");
    if (!kLab) {
        const int til = (p->variant >> 1) & 7;
        if ((p->variant & ~kProductVariantMask) || !(til == 0 || til == 1 || til == 4 || til == 7))
            return fail(VATTN_K_ERR_INVALID, "variant selects a measurement build that this library does not contain (tools/lab/libvattn_lab.so, -DVATTN_LAB)");
    }
    return VATTN_K_OK;
}

}  // namespace vattn_k

using namespace vattn_k;

extern "C" {

const char* vattn_kernels_last_error(void) { return g_err.c_str(); }

size_t vattn_attn_workspace_bytes(const vattn_attn_params* p) {
    if (!abi_ok(p)) return 0;
    if (!p || p->h_k <= 0 || p->h <= 0 || p->b <= 0 || p->seqlen_q <= 0) return 0;
    return p->seqlen_q != 1 ? prefill_workspace_bytes(p) : decode_workspace_bytes(p);
}

int vattn_flash_attn_with_kvcache(const vattn_attn_params* p, void* stream) {
    int rc = validate(p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (p->k_new && p->seqlen_knew > 0 && !p->cache_seqlens) return fail(VATTN_K_ERR_INVALID, "If key is supplied, seqlens_k must also be passed in");
    return p->seqlen_q == 1 ? launch_decode_form(p, st) : launch_prefill_form(p, st);
}

int vattn_attn_plan_describe(const vattn_attn_params* p, vattn_plan_desc* out) {
    if (!abi_ok(p)) return fail(VATTN_K_ERR_INVALID, "vattn_attn_params: struct_size / abi_version do not match this library");
    if (!p || !out || p->h_k <= 0 || p->h <= 0 || p->b <= 0 || p->seqlen_q <= 0 || (p->d != 64 && p->d != 128)) return fail(VATTN_K_ERR_INVALID, "vattn_attn_plan_describe: bad shape");
    memset(out, 0, sizeof *out);
    if (p->seqlen_q == 1) decode_describe(p, out);
    else prefill_describe(p, out);
    out->workspace_bytes = (int64_t)vattn_attn_workspace_bytes(p);
    return VATTN_K_OK;
}

int32_t vattn_decode_plan(const vattn_attn_params* p, const int32_t* cache_seqlens_host, vattn_decode_item* items_out, int32_t cap, int32_t* seq_out) {
    if (!abi_ok(p)) return VATTN_K_ERR_INVALID;
    return decode_plan(p, cache_seqlens_host, items_out, cap, seq_out);
}

int32_t vattn_prefill_plan(const vattn_attn_params* p, const int32_t* q_lens_host, const int32_t* k_lens_host, vattn_prefill_item* items_out,
                           int32_t cap_items, vattn_prefill_item* blocks_out, int32_t cap_blocks, int32_t* counts_out) {
    if (!abi_ok(p)) return VATTN_K_ERR_INVALID;
    return prefill_worklist(p, q_lens_host, k_lens_host, items_out, cap_items, blocks_out, cap_blocks, counts_out);
}

int32_t vattn_prefill_plan_wg(const vattn_attn_params* p, const int32_t* q_lens_host, const int32_t* k_lens_host, vattn_prefill_item* items_out,
                              int32_t cap_items, vattn_prefill_item* blocks_out, int32_t cap_blocks, int32_t* wg_first_out, int32_t max_wg,
                              int32_t* counts_out) {
    if (!abi_ok(p)) return VATTN_K_ERR_INVALID;
    return prefill_worklist(p, q_lens_host, k_lens_host, items_out, cap_items, blocks_out, cap_blocks, counts_out, wg_first_out, max_wg, wg_first_out ? 1 : 2);
}

size_t vattn_hybrid_workspace_bytes(const vattn_attn_params* prefill, const vattn_attn_params* decode) {
    if (!abi_ok(prefill) || !abi_ok(decode)) return 0;
    if (!prefill || !decode || decode->h_k <= 0 || decode->h <= 0 || decode->b <= 0) return 0;
    return hybrid_workspace_bytes(prefill, decode);
}

int vattn_hybrid_attn(const vattn_attn_params* prefill, const vattn_attn_params* decode, void* workspace, void* stream) {
    int rc = validate(prefill);
    if (rc) return rc;
    rc = validate(decode);
    if (rc) return rc;
    return launch_hybrid(prefill, decode, workspace, (hipStream_t)stream);
}

int vattn_selftest_layouts(void* stream, int32_t* detail_out) {
    hipStream_t st = (hipStream_t)stream;
    int* d = nullptr;
    if (hipMalloc(&d, 8 * sizeof(int)) != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, "hipMalloc failed");
    (void)hipMemsetAsync(d, 0, 8 * sizeof(int), st);      // (a failure of any of these surfaces in the synchronising copy below)
    unsigned* gsrc = nullptr;
    unsigned hsrc[512];
    for (int i = 0; i < 512; i++) hsrc[i] = (unsigned)i;
    if (hipMalloc(&gsrc, sizeof(hsrc)) != hipSuccess) { (void)hipFree(d); return fail(VATTN_K_ERR_LAUNCH, "hipMalloc failed"); }
    (void)hipMemcpyAsync(gsrc, hsrc, sizeof(hsrc), hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, d, (const unsigned*)gsrc);
    (void)hipFuncSetAttribute((const void*)selftest_dma_high_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    hipLaunchKernelGGL(selftest_dma_high_kernel, dim3(1), dim3(64), 96 << 10, st, d, (const unsigned*)gsrc);
    int h[8] = {0};
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    (void)hipFree(gsrc);
    if (e != hipSuccess) return fail(VATTN_K_ERR_LAUNCH, hipGetErrorString(e));
    int bad = 0;
    for (int i = 0; i < 8; i++) {
        if (detail_out) detail_out[i] = h[i];
        if (i < 6) bad |= h[i];            // [6], [7] report LDS-DMA behaviour (out-of-range lanes, destinations beyond 64 KiB)
    }
    return bad ? fail(VATTN_K_ERR_INVALID, "hardware layout assumption violated") : VATTN_K_OK;
}

float vattn_time_attn(const vattn_attn_params* p, void* stream, int32_t warmup, int32_t iters) {
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < warmup; i++)
        if (vattn_flash_attn_with_kvcache(p, stream) != 0) return -1.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return -1.f;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return -1.f; }
    hipError_t e = hipEventRecord(e0, st);
    for (int i = 0; i < iters && e == hipSuccess; i++)
        if (vattn_flash_attn_with_kvcache(p, stream) != 0) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return -1.f;
    return ms / (iters > 0 ? iters : 1);
}

}  // extern "C"
