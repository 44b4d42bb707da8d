// HIP virtual-memory backend of the page manager: the product path on MI355X.
// Replaces the reference's CUDA-driver backend (/root/reference/vattention/cudaInternal.h:15-94,
// vtensor.h:21-46) and, for sub-2MiB pages, its patched-UVM-driver backend (uvmInternal.h) — on
// gfx950 small pages are just a smaller multiple of the VMM granularity, no driver patch.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "page_manager.h"

namespace vattn {

struct HipCtx {
    int device;
    hipMemAllocationProp prop;
    hipMemAccessDesc access;
    std::vector<void*> flush_allocs;     // see h_tlb_flush
    std::vector<hipEvent_t> fences;      // per request slot: recorded by free_batch_idx_on_stream, see h_fence_*
    std::vector<uint8_t> fence_set;
    std::mutex mu;
};
// device memory the TLB-invalidation probes may hold at any time (kFlushPark parked 2 MiB allocations); reserve_physical_pages
// sizes the pool against free memory MINUS this, so the probe's hipMalloc cannot fail because the KV pool took everything
constexpr size_t kFlushBytes = 2u << 20;
constexpr size_t kFlushPark = 32;

static int hip_fail(const char* what, hipError_t e) {
    fprintf(stderr, "[vattn] %s failed: %s (%d)\n", what, hipGetErrorString(e), (int)e);
    return -1;
}

static int h_thread_init(void* c) {
    auto* x = (HipCtx*)c;
    hipError_t e = hipSetDevice(x->device);
    return e == hipSuccess ? 0 : hip_fail("hipSetDevice", e);
}
static int h_granularity(void* c, uint64_t* mn, uint64_t* rec) {
    auto* x = (HipCtx*)c;
    size_t a = 0, b = 0;
    hipError_t e = hipMemGetAllocationGranularity(&a, &x->prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return hip_fail("hipMemGetAllocationGranularity(min)", e);
    e = hipMemGetAllocationGranularity(&b, &x->prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return hip_fail("hipMemGetAllocationGranularity(rec)", e);
    *mn = a;
    *rec = b;
    return 0;
}
static int h_reserve(void*, uint64_t bytes, uint64_t align, uint64_t* out) {
    void* p = nullptr;
    hipError_t e = hipMemAddressReserve(&p, bytes, align, nullptr, 0);
    if (e != hipSuccess) return hip_fail("hipMemAddressReserve", e);
    *out = (uint64_t)p;
    return 0;
}
static int h_free_va(void*, uint64_t base, uint64_t bytes) {
    hipError_t e = hipMemAddressFree((void*)base, bytes);
    return e == hipSuccess ? 0 : hip_fail("hipMemAddressFree", e);
}
static int h_create(void* c, uint64_t bytes, uint64_t* out) {
    auto* x = (HipCtx*)c;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, bytes, &x->prop, 0);
    if (e != hipSuccess) return hip_fail("hipMemCreate", e);
    static_assert(sizeof(h) <= sizeof(uint64_t), "handle must fit in 64 bits");
    uint64_t v = 0;
    memcpy(&v, &h, sizeof(h));
    *out = v;
    return 0;
}
static int h_release(void*, uint64_t handle) {
    hipMemGenericAllocationHandle_t h;
    memcpy(&h, &handle, sizeof(h));
    hipError_t e = hipMemRelease(h);
    return e == hipSuccess ? 0 : hip_fail("hipMemRelease", e);
}
static int h_map(void*, uint64_t va, uint64_t bytes, uint64_t handle) {
    hipMemGenericAllocationHandle_t h;
    memcpy(&h, &handle, sizeof(h));
    hipError_t e = hipMemMap((void*)va, bytes, 0, h, 0);   // offset "currently must be zero" (hip_runtime_api.h:9396-9407)
    return e == hipSuccess ? 0 : hip_fail("hipMemMap", e);
}
static int h_access(void* c, uint64_t va, uint64_t bytes) {
    auto* x = (HipCtx*)c;
    hipError_t e = hipMemSetAccess((void*)va, bytes, &x->access, 1);
    return e == hipSuccess ? 0 : hip_fail("hipMemSetAccess", e);
}
static int h_unmap(void*, uint64_t va, uint64_t bytes) {
    hipError_t e = hipMemUnmap((void*)va, bytes);
    return e == hipSuccess ? 0 : hip_fail("hipMemUnmap", e);
}

// TLB invalidation after unmap.  Measured on MI355X / ROCm 7.2 (tools/remap_probe3.cpp, remap_probe4.cpp):
// after hipMemUnmap + hipMemMap(other handle) at the same address a kernel still reads the OLD physical
// page — sleeping, hipDeviceSynchronize or launching kernels do not help — until the driver services an
// ordinary device allocation (its map ioctl carries the pending invalidation).  A 2 MiB hipMalloc costs
// ~160 us and is only paid by batches that unmapped something.  hipFree synchronises the device, so the
// probe allocations are parked and released in bulk (every 32 flushes and at teardown).
static int h_tlb_flush(void* c) {
    auto* x = (HipCtx*)c;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, kFlushBytes);
    if (e != hipSuccess) {
        // memory pressure: give the parked probes back and try once more (their budget is excluded from the KV pool, h_mem_info)
        (void)hipGetLastError();
        std::vector<void*> parked;
        {
            std::lock_guard<std::mutex> l(x->mu);
            parked.swap(x->flush_allocs);
        }
        for (void* q : parked) (void)hipFree(q);
        e = hipMalloc(&p, kFlushBytes);
        if (e != hipSuccess) return hip_fail("hipMalloc (TLB invalidation probe)", e);
    }
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> l(x->mu);
        x->flush_allocs.push_back(p);
        if (x->flush_allocs.size() >= kFlushPark) drop.swap(x->flush_allocs);
    }
    for (void* q : drop) (void)hipFree(q);     // hipFree synchronises the device: once per kFlushPark unmapping batches
    return 0;
}

int hip_versions(int* rt, int* drv) {      // include/vattn.h vattn_hip_versions
    int a = 0, b = 0;
    if (hipRuntimeGetVersion(&a) != hipSuccess || hipDriverGetVersion(&b) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (rt) *rt = a;
    if (drv) *drv = b;
    return 0;
}

static int h_quiesce(void*) {
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : hip_fail("hipDeviceSynchronize", e);
}

static int h_mem_info(void* c, uint64_t* free_b, uint64_t* total_b) {
    (void)c;
    size_t f = 0, t = 0;
    hipError_t e = hipMemGetInfo(&f, &t);
    if (e != hipSuccess) return hip_fail("hipMemGetInfo", e);
    const size_t budget = kFlushBytes * kFlushPark;
    *free_b = f > budget ? f - budget : 0;
    *total_b = t;
    return 0;
}

// Per-slot fences.  The engine frees a slot right after LAUNCHING the iteration that last reads its pages; an event recorded
// on that stream at free time marks the point after which the pages may be unmapped.  Waiting for it replaces the
// device-wide synchronisation before unmaps whenever the engine goes through vattn_free_batch_idx_on_stream.
static int h_fence_record(void* c, uint32_t slot, void* stream) {
    auto* x = (HipCtx*)c;
    std::lock_guard<std::mutex> l(x->mu);
    if (x->fences.size() <= slot) {
        x->fences.resize((size_t)slot + 1, nullptr);
        x->fence_set.resize((size_t)slot + 1, 0);
    }
    if (!stream && !x->fence_set[slot]) return 0;
    x->fence_set[slot] = 0;
    if (!stream) return 0;                     // plain free: forget the fence (a reclaim falls back to quiesce)
    if (!x->fences[slot]) {
        hipError_t e = hipEventCreateWithFlags(&x->fences[slot], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail("hipEventCreateWithFlags", e);
    }
    // (void*)-1 stands for the legacy default stream (NULL means "no fence" in the ops table)
    hipStream_t st = stream == (void*)-1 ? nullptr : (hipStream_t)stream;
    hipError_t e = hipEventRecord(x->fences[slot], st);
    if (e != hipSuccess) return hip_fail("hipEventRecord", e);
    x->fence_set[slot] = 1;
    return 0;
}
static int h_fence_wait(void* c, uint32_t slot) {
    auto* x = (HipCtx*)c;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> l(x->mu);
        if (slot >= x->fences.size() || !x->fence_set[slot]) return 1;
        ev = x->fences[slot];
    }
    hipError_t e = hipEventSynchronize(ev);
    return e == hipSuccess ? 0 : hip_fail("hipEventSynchronize", e);
}

// Fills `ops` with the HIP VMM table for `device`; the context object lives for the process.
int make_hip_backend(int device, vattn_backend_ops* ops) {
    // A HIP context must exist on the calling thread ("initialize PyTorch first", cudaInternal.h:20-25);
    // hipSetDevice creates/binds the primary context, so no torch dependency is needed here.
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    auto* x = new HipCtx();
    x->device = device;
    memset(&x->prop, 0, sizeof(x->prop));
    x->prop.type = hipMemAllocationTypePinned;
    x->prop.location.type = hipMemLocationTypeDevice;
    x->prop.location.id = device;
    memset(&x->access, 0, sizeof(x->access));
    x->access.location.type = hipMemLocationTypeDevice;
    x->access.location.id = device;
    x->access.flags = hipMemAccessFlagsProtReadWrite;
    ops->ctx = x;
    ops->granularity = h_granularity;
    ops->reserve_va = h_reserve;
    ops->free_va = h_free_va;
    ops->create = h_create;
    ops->release = h_release;
    ops->map = h_map;
    ops->set_access = h_access;
    ops->unmap = h_unmap;
    ops->thread_init = h_thread_init;
    ops->tlb_flush = h_tlb_flush;
    ops->quiesce = h_quiesce;
    ops->mem_info = h_mem_info;
    ops->fence_record = h_fence_record;
    ops->fence_wait = h_fence_wait;
    return 0;
}

}  // namespace vattn
