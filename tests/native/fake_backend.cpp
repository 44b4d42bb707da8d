// Host-only physical backend for driving libvattn_amd.so's bookkeeping core on a CPU-only box.
// TEST INFRASTRUCTURE ONLY (built into tests/native/libvattn_fake_backend.so by __graft_entry__.build()).
// It keeps a model of the device's VMM state and counts contract violations, so tests can assert
// that the manager never double-maps, never unmaps what is not mapped, never sets access on a hole.
#include <cstdint>
#include <map>
#include <mutex>
#include <set>

#include "../../include/vattn.h"

namespace {
struct Fake {
    std::mutex mu;
    uint64_t next_va = 0x7f0000000000ull;
    uint64_t next_handle = 1;
    uint64_t min_gran = 4096, rec_gran = 2ull << 20;
    std::map<uint64_t, uint64_t> reserved;                 // base -> bytes
    std::map<uint64_t, std::pair<uint64_t, uint64_t>> mapped;   // va -> (bytes, handle)
    std::set<uint64_t> live_handles;
    std::set<uint64_t> accessible;                          // va of mapped pages with access set
    std::set<uint64_t> stale;                               // VAs unmapped since the last TLB flush (GPU may still translate them)
    uint64_t n_flush = 0, n_quiesce = 0;
    bool quiesced_since_map = true;     // set by quiesce, cleared by map: an unmap must see it set (or no map since)
    uint64_t violations = 0, n_create = 0, n_map = 0, n_access = 0, n_unmap = 0, n_release = 0;
    uint64_t fail_create_after = ~0ull, fail_map_after = ~0ull;
    std::set<uint32_t> fences;                              // slots with a recorded fence
    uint64_t n_fence_wait = 0;
    int delay_us = 0;
    bool validate = true;      // false: count calls only (tools/pagemgr_steps_bench.py prices the manager, not this double's containers)
};
Fake g;

int f_gran(void*, uint64_t* a, uint64_t* b) { *a = g.min_gran; *b = g.rec_gran; return 0; }
int f_reserve(void*, uint64_t bytes, uint64_t align, uint64_t* out) {
    std::lock_guard<std::mutex> l(g.mu);
    uint64_t p = ((g.next_va + align - 1) / align) * align;
    g.next_va = p + bytes + align;
    g.reserved[p] = bytes;
    *out = p;
    return 0;
}
int f_free_va(void*, uint64_t base, uint64_t bytes) {
    std::lock_guard<std::mutex> l(g.mu);
    auto it = g.reserved.find(base);
    if (it == g.reserved.end() || it->second != bytes) { g.violations++; return -1; }
    for (auto& kv : g.mapped)
        if (kv.first >= base && kv.first < base + bytes) { g.violations++; break; }   // freeing VA with live mappings
    g.reserved.erase(it);
    return 0;
}
int f_create(void*, uint64_t, uint64_t* out) {
    std::lock_guard<std::mutex> l(g.mu);
    if (g.n_create >= g.fail_create_after) return -1;
    g.n_create++;
    *out = g.next_handle++;
    if (g.validate) g.live_handles.insert(*out);
    return 0;
}
int f_release(void*, uint64_t h) {
    std::lock_guard<std::mutex> l(g.mu);
    if (!g.validate) { g.n_release++; return 0; }
    if (!g.live_handles.erase(h)) { g.violations++; return -1; }
    g.n_release++;
    return 0;
}
bool inside_reservation(uint64_t va, uint64_t bytes) {
    auto it = g.reserved.upper_bound(va);
    if (it == g.reserved.begin()) return false;
    --it;
    return va >= it->first && va + bytes <= it->first + it->second;
}
int f_map(void*, uint64_t va, uint64_t bytes, uint64_t h) {
    std::lock_guard<std::mutex> l(g.mu);
    if (g.n_map >= g.fail_map_after) return -1;            // injected driver failure (not a contract violation)
    g.n_map++;
    if (!g.validate) return 0;
    if (!inside_reservation(va, bytes) || !g.live_handles.count(h) || g.mapped.count(va) || va % g.min_gran) { g.violations++; return -1; }
    g.mapped[va] = {bytes, h};
    return 0;
}
int f_access(void*, uint64_t va, uint64_t bytes) {
    std::lock_guard<std::mutex> l(g.mu);
    g.n_access++;
    if (!g.validate) return 0;
    uint64_t p = va;
    while (p < va + bytes) {            // the range must be exactly covered by whole mappings
        auto it = g.mapped.find(p);
        if (it == g.mapped.end()) { g.violations++; return -1; }
        g.accessible.insert(p);
        p += it->second.first;
    }
    if (p != va + bytes) { g.violations++; return -1; }
    return 0;
}
int f_unmap(void*, uint64_t va, uint64_t bytes) {
    std::lock_guard<std::mutex> l(g.mu);
    g.n_unmap++;
    if (!g.validate) return 0;
    // (unmaps of a freed slot's pages must follow a fence wait or a quiesce; pages an ACTIVE slot no longer needs may go at once)
    auto it = g.mapped.find(va);
    if (it == g.mapped.end() || it->second.first != bytes) { g.violations++; return -1; }
    g.mapped.erase(it);
    g.accessible.erase(va);
    g.stale.insert(va);
    return 0;
}
int f_quiesce(void*) {
    std::lock_guard<std::mutex> l(g.mu);
    g.n_quiesce++;
    g.quiesced_since_map = true;
    return 0;
}
int f_flush(void*) {
    std::lock_guard<std::mutex> l(g.mu);
    g.n_flush++;
    g.stale.clear();
    return 0;
}
int f_fence_record(void*, uint32_t slot, void* stream) {
    std::lock_guard<std::mutex> l(g.mu);
    if (stream) g.fences.insert(slot); else g.fences.erase(slot);
    return 0;
}
int f_fence_wait(void*, uint32_t slot) {
    std::lock_guard<std::mutex> l(g.mu);
    if (!g.fences.count(slot)) return 1;
    g.n_fence_wait++;
    return 0;
}
vattn_backend_ops g_ops = {nullptr, f_gran, f_reserve, f_free_va, f_create, f_release, f_map, f_access, f_unmap, nullptr, f_flush, f_quiesce,
                           nullptr, f_fence_record, f_fence_wait};
}  // namespace

extern "C" {
const vattn_backend_ops* vattn_fake_backend_ops() { return &g_ops; }
void vattn_fake_reset(uint64_t min_gran, uint64_t rec_gran) {
    std::lock_guard<std::mutex> l(g.mu);
    g.reserved.clear(); g.mapped.clear(); g.live_handles.clear(); g.accessible.clear(); g.stale.clear(); g.n_flush = 0; g.n_quiesce = 0;
    g.violations = g.n_create = g.n_map = g.n_access = g.n_unmap = g.n_release = 0;
    g.next_handle = 1; g.next_va = 0x7f0000000000ull; g.fail_create_after = ~0ull; g.fail_map_after = ~0ull;
    g.fences.clear(); g.n_fence_wait = 0;
    g.min_gran = min_gran; g.rec_gran = rec_gran; g.validate = true;
}
void vattn_fake_set_validate(int on) { g.validate = on != 0; }
void vattn_fake_fail_create_after(uint64_t n) { g.fail_create_after = n; }
void vattn_fake_fail_map_after(uint64_t n) { g.fail_map_after = n; }     // the (n+1)-th map call from now on the counter fails
uint64_t vattn_fake_quiesce_count() { return g.n_quiesce; }
uint64_t vattn_fake_fence_wait_count() { return g.n_fence_wait; }
// out = [violations, n_create, n_map, n_access, n_unmap, n_release, live_handles, mapped_pages, accessible_pages, reserved_ranges,
//        n_flush, stale_vas (unmapped since the last flush)]
void vattn_fake_counters(uint64_t* out) {
    std::lock_guard<std::mutex> l(g.mu);
    out[0] = g.violations; out[1] = g.n_create; out[2] = g.n_map; out[3] = g.n_access; out[4] = g.n_unmap;
    out[5] = g.n_release; out[6] = g.live_handles.size(); out[7] = g.mapped.size(); out[8] = g.accessible.size();
    out[9] = g.reserved.size();
    out[10] = g.n_flush; out[11] = g.stale.size();
}
// every mapped page as (va, bytes, handle, accessible) rows; returns rows or -needed
int64_t vattn_fake_mapped(uint64_t* out, uint64_t cap_rows) {
    std::lock_guard<std::mutex> l(g.mu);
    if (g.mapped.size() > cap_rows) return -(int64_t)g.mapped.size();
    uint64_t n = 0;
    for (auto& kv : g.mapped) {
        out[4 * n] = kv.first; out[4 * n + 1] = kv.second.first; out[4 * n + 2] = kv.second.second;
        out[4 * n + 3] = g.accessible.count(kv.first);
        n++;
    }
    return (int64_t)n;
}
}
