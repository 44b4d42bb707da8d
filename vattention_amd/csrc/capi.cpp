// extern "C" surface of libvattn_amd.so for the page manager (include/vattn.h).
#include <map>
#include <mutex>
#include <new>

#include "page_manager.h"

namespace vattn {
int make_hip_backend(int device, vattn_backend_ops* ops);
int hip_vmm_selfcheck(int device, const vattn_backend_ops* ops, uint32_t detail[3]);   // vmm_selfcheck.hip
int hip_versions(int* rt, int* drv);                                                       // hip_backend.cpp
}

struct vattn_handle {
    vattn::PageManager* pm;
    std::string create_error;
};

static thread_local std::string g_create_error;

extern "C" {

int vattn_create(const vattn_config* cfg, const vattn_backend_ops* backend, vattn_t** out) {
    if (!cfg || !out) return VATTN_ERR_INVALID;
    *out = nullptr;
    vattn_backend_ops ops;
    if (backend) {
        ops = *backend;
    } else {
        if (vattn::make_hip_backend(cfg->device, &ops) != 0) return VATTN_ERR_DRIVER;   // no silent fallback
    }
    auto* h = new (std::nothrow) vattn_handle();
    if (!h) return VATTN_ERR_INVALID;
    h->pm = new vattn::PageManager(*cfg, ops);
    if (!backend && !(cfg->flags & VATTN_FLAG_NO_VMM_SELFCHECK)) {
        // Unmap safety rests on the backend's TLB-invalidation policy (hip_backend.cpp): prove once per device and process
        // that a kernel sees the NEW page after unmap + map at the same address, and refuse to serve otherwise.
        static std::mutex mu;
        static std::map<int, int> verdict;
        std::lock_guard<std::mutex> l(mu);
        auto it = verdict.find(cfg->device);
        if (it == verdict.end()) it = verdict.emplace(cfg->device, vattn::hip_vmm_selfcheck(cfg->device, &ops, nullptr)).first;
        if (it->second != 0) {
            int rt = 0, drv = 0;
            (void)vattn::hip_versions(&rt, &drv);
            const std::string ver = " [HIP runtime " + std::to_string(rt) + ", driver " + std::to_string(drv) + "]";
            h->pm->set_error((it->second > 0
                ? std::string("HIP VMM self-check failed: a kernel still reads the OLD physical page after hipMemUnmap + hipMemMap at the same "
                  "virtual address, even after the TLB-invalidation step — refusing to start (reclaimed KV pages would leak between "
                  "requests).  The step was verified on ROCm 7.2 / gfx950 and is re-proven at every start; set VATTN_FLAG_NO_VMM_SELFCHECK only to debug.")
                : std::string("HIP VMM self-check could not run (driver error)")) + ver);
            *out = h;
            return VATTN_ERR_DRIVER;
        }
    }
    int rc = h->pm->init();
    *out = h;            // returned even on failure so the caller can read vattn_last_error()
    return rc;
}

int vattn_num_tensors(const vattn_t* m) { return m->pm->num_tensors(); }
uint64_t vattn_tensor_base(const vattn_t* m, int i) {
    return (i < 0 || i >= m->pm->num_tensors()) ? 0 : m->pm->tensor_base(i);
}
int vattn_get_layout(const vattn_t* m, vattn_layout* out) { m->pm->layout(out); return VATTN_OK; }
int64_t vattn_reserve_physical_pages(vattn_t* m, uint64_t free_memory) { return m->pm->reserve_physical_pages(free_memory); }
int vattn_step(vattn_t* m, const uint64_t* l, uint32_t n, int eager) { return m->pm->step(l, n, eager != 0); }
int vattn_step_async(vattn_t* m, const uint64_t* l, uint32_t n) { return m->pm->step_async(l, n); }
int vattn_wait(vattn_t* m) { return m->pm->wait(); }
int vattn_alloc_new_batch_idx(vattn_t* m, uint64_t seqlen) { return m->pm->alloc_new_batch_idx(seqlen); }
int vattn_free_batch_idx(vattn_t* m, int slot) { return m->pm->free_batch_idx(slot); }
int vattn_free_batch_idx_on_stream(vattn_t* m, int slot, void* stream) { return m->pm->free_batch_idx(slot, stream, true); }
int vattn_premap(vattn_t* m, uint64_t seqlen) { return m->pm->premap(seqlen); }
int vattn_cancel_premap(vattn_t* m, int slot) { return m->pm->cancel_premap(slot); }
int64_t vattn_wait_pool_ready(vattn_t* m, int64_t timeout_ms) { return m->pm->wait_pool_ready(timeout_ms); }
int vattn_wait_layer(vattn_t* m, uint32_t layer) { return m->pm->wait_layer(layer); }
uint32_t vattn_layers_ready(vattn_t* m) { return m->pm->layers_ready(); }
int vattn_set_sync_layers(vattn_t* m, uint32_t n) { return m->pm->set_sync_layers(n); }
uint64_t vattn_num_free_kvblocks(vattn_t* m) { return m->pm->num_free_kvblocks(); }
int vattn_set_deferred_reclamation(vattn_t* m, int on) { return m->pm->set_deferred_reclamation(on != 0); }
int vattn_set_verbose(vattn_t* m, int on) { return m->pm->set_verbose(on != 0); }
int vattn_map_common_pages(vattn_t* m, uint64_t n) { return m->pm->map_common_pages(n); }
int vattn_show_kvcache_config(vattn_t* m) { return m->pm->show_kvcache_config(); }
int vattn_show_allocator_state(vattn_t* m) { return m->pm->show_allocator_state(); }
int vattn_cleanup(vattn_t* m) { return m->pm->cleanup(); }
void vattn_destroy(vattn_t* m) {
    if (!m) return;
    delete m->pm;
    delete m;
}
int64_t vattn_state_dump(vattn_t* m, uint64_t* out, uint64_t cap) { return m->pm->state_dump(out, cap); }
int64_t vattn_pagemap_dump(vattn_t* m, uint64_t* out, uint64_t cap_rows) { return m->pm->pagemap_dump(out, cap_rows); }
int vattn_get_stats(vattn_t* m, vattn_stats* out) { m->pm->stats(out); return VATTN_OK; }
int vattn_get_counts(vattn_t* m, uint64_t out[4]) { m->pm->counts(out); return VATTN_OK; }
const char* vattn_last_error(const vattn_t* m) { return m ? m->pm->last_error() : "null handle"; }

int vattn_vmm_selfcheck(int device, uint32_t detail[3]) {
    vattn_backend_ops ops;
    if (vattn::make_hip_backend(device, &ops) != 0) return VATTN_ERR_DRIVER;
    return vattn::hip_vmm_selfcheck(device, &ops, detail);
}

int vattn_hip_versions(int* rt, int* drv) { return vattn::hip_versions(rt, drv) == 0 ? VATTN_OK : VATTN_ERR_DRIVER; }

int vattn_hip_granularity(int device, uint64_t* mn, uint64_t* rec) {
    vattn_backend_ops ops;
    if (vattn::make_hip_backend(device, &ops) != 0) return VATTN_ERR_DRIVER;
    return ops.granularity(ops.ctx, mn, rec) == 0 ? VATTN_OK : VATTN_ERR_DRIVER;
}

}  // extern "C"
