// Start-up self-check of the HIP VMM backend's unmap policy (include/vattn.h: vattn_vmm_selfcheck).
//
// vAttention moves physical pages between request slots: unmap from one virtual range, map under another.  On ROCm 7.2 /
// gfx950 a kernel keeps translating an unmapped-and-remapped virtual page to the OLD physical page until the driver services an
// ordinary allocation (tools/remap_probe3.cpp, remap_probe4.cpp); the backend's tlb_flush op exists for exactly that
// (hip_backend.cpp) and is undocumented behaviour, so it is PROVEN here once per device before the manager serves anything
// (role of the reference's driver-error checks, /root/reference/vattention/cudaInternal.h:1-13,70-94 — it never remaps under
// a live translation without cuMemUnmap's own invalidation).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vattn.h"

namespace vattn {

namespace {
__global__ void fill_kernel(uint32_t* p, uint32_t v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// reads every word (so every translation of the range is exercised) and reports the first and the OR of mismatches
__global__ void read_kernel(const uint32_t* p, size_t n, uint32_t* out) {
    uint32_t first = p[0];
    uint32_t diff = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) diff |= p[i] ^ first;
    if (diff) atomicOr(&out[1], diff);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = first;
}
}  // namespace

int hip_vmm_selfcheck(int device, const vattn_backend_ops* be, uint32_t detail[3]) {
    if (hipSetDevice(device) != hipSuccess) return VATTN_ERR_DRIVER;
    uint64_t mn = 0, rec = 0;
    if (be->granularity(be->ctx, &mn, &rec) != 0 || mn == 0) return VATTN_ERR_DRIVER;
    const uint64_t bytes = (2ull << 20) % mn == 0 ? (2ull << 20) : mn;
    const size_t words = bytes / 4;
    uint64_t va = 0, va2 = 0, ha = 0, hb = 0;
    uint32_t* dres = nullptr;
    hipStream_t st = nullptr;
    int rc = VATTN_ERR_DRIVER;
    uint32_t h[2] = {0, 0}, before = 0, no_flush = 0, after = 0;
    bool mapped1 = false, mapped2 = false;
    auto read_back = [&](uint64_t at, uint32_t* first) -> bool {
        if (hipMemsetAsync(dres, 0, 8, st) != hipSuccess) return false;
        hipLaunchKernelGGL(read_kernel, dim3(64), dim3(256), 0, st, (const uint32_t*)at, words, dres);
        if (hipMemcpyAsync(h, dres, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        *first = h[0];
        return true;
    };
    do {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        if (hipMalloc(&dres, 8) != hipSuccess) break;
        if (be->reserve_va(be->ctx, bytes, rec > bytes ? rec : bytes, &va) != 0) break;
        if (be->reserve_va(be->ctx, bytes, rec > bytes ? rec : bytes, &va2) != 0) break;
        if (be->create(be->ctx, bytes, &ha) != 0 || be->create(be->ctx, bytes, &hb) != 0) break;
        // page B gets its pattern through a second address
        if (be->map(be->ctx, va2, bytes, hb) != 0) break;
        mapped2 = true;
        if (be->set_access(be->ctx, va2, bytes) != 0) break;
        hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, (uint32_t*)va2, 0xB0B0B0B0u, words);
        // page A at the address under test: written and read by kernels (translations now cached)
        if (be->map(be->ctx, va, bytes, ha) != 0) break;
        mapped1 = true;
        if (be->set_access(be->ctx, va, bytes) != 0) break;
        hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, (uint32_t*)va, 0xA0A0A0A0u, words);
        if (!read_back(va, &before)) break;
        // move the address to page B, exactly as a reclaim + re-map does
        if (be->unmap(be->ctx, va, bytes) != 0) break;
        mapped1 = false;
        if (be->map(be->ctx, va, bytes, hb) != 0) break;
        mapped1 = true;
        if (be->set_access(be->ctx, va, bytes) != 0) break;
        if (!read_back(va, &no_flush)) break;               // informational: what a kernel sees WITHOUT the policy
        if (be->tlb_flush && be->tlb_flush(be->ctx) != 0) break;
        if (!read_back(va, &after)) break;
        rc = (after == 0xB0B0B0B0u && h[1] == 0) ? 0 : 1;
    } while (0);
    if (detail) { detail[0] = before; detail[1] = no_flush; detail[2] = after; }
    if (st) (void)hipStreamSynchronize(st);
    if (mapped1) be->unmap(be->ctx, va, bytes);
    if (mapped2) be->unmap(be->ctx, va2, bytes);
    if (ha) be->release(be->ctx, ha);
    if (hb) be->release(be->ctx, hb);
    if (va) be->free_va(be->ctx, va, bytes);
    if (va2) be->free_va(be->ctx, va2, bytes);
    if (dres) (void)hipFree(dres);
    if (st) (void)hipStreamDestroy(st);
    if (be->tlb_flush) (void)be->tlb_flush(be->ctx);        // the unmaps above leave stale entries too
    (void)hipGetLastError();
    return rc;
}

}  // namespace vattn
