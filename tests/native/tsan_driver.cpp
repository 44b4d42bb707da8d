// ThreadSanitizer driver for the page manager's mapper thread (the reference's background thread mutates
// shared vectors unlocked; this design must be race-free).  Built with g++ -fsanitize=thread from page_manager.cpp +
// capi.cpp + fake_backend.cpp (no HIP).  The API thread hammers step_async / alloc / free / num_free_kvblocks while the
// mapper thread executes look-ahead batches and pre-creates handles; exits non-zero on any mismatch, TSan reports races.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/vattn.h"

extern "C" const vattn_backend_ops* vattn_fake_backend_ops();
extern "C" void vattn_fake_reset(uint64_t, uint64_t);
extern "C" void vattn_fake_counters(uint64_t*);
namespace vattn {
int make_hip_backend(int, vattn_backend_ops*) { return -1; }   // never used: a backend table is always passed
int hip_vmm_selfcheck(int, const vattn_backend_ops*, uint32_t*) { return -1; }
int hip_versions(int*, int*) { return -1; }
}

int main() {
    vattn_fake_reset(4096, 2 << 20);
    vattn_config cfg = {};
    cfg.num_layers = 4; cfg.num_kv_heads = 2; cfg.head_size = 128; cfg.max_batch_size = 16; cfg.max_context_length = 4096;
    cfg.itemsize = 2; cfg.device = 0; cfg.page_size = 64 << 10; cfg.megacache = 0;
    cfg.flags = VATTN_FLAG_LAYERED_ASYNC;     // new prompts are mapped layer-ordered by the mapper while this thread polls wait_layer
    vattn_t* m = nullptr;
    if (vattn_create(&cfg, vattn_fake_backend_ops(), &m) != 0) { printf("create failed: %s\n", vattn_last_error(m)); return 2; }
    vattn_reserve_physical_pages(m, 600ull * 8 * (64 << 10));
    std::mt19937 rng(7);
    std::vector<uint64_t> lens(16, 0);
    std::vector<uint64_t> target(16, 0);
    int pre_slot = -1;
    uint64_t pre_len = 0;
    for (int it = 0; it < 4000; it++) {
        if (pre_slot >= 0) {                                     // the request looked ahead for arrives (or is dropped)
            if (rng() % 8 == 0) vattn_cancel_premap(m, pre_slot);
            else { lens[pre_slot] = pre_len; target[pre_slot] = pre_len + rng() % 300; }
            pre_slot = -1;
        }
        if (rng() % 3 == 0) {
            const uint64_t n = 1 + rng() % 2000;
            const int s = vattn_alloc_new_batch_idx(m, n);
            if (s >= 0) { lens[s] = n; target[s] = n + rng() % 300; }
        }
        const int rc = (it % 5 == 4) ? vattn_step(m, lens.data(), 16, 1) : vattn_step_async(m, lens.data(), 16);
        if (rc != 0 && rc != VATTN_ERR_OOM) { printf("step failed %d: %s\n", rc, vattn_last_error(m)); return 3; }
        (void)vattn_num_free_kvblocks(m);
        if (rng() % 4 == 0) {                                    // admission look-ahead while the mapper still works on the step's batch
            pre_len = 1 + rng() % 2000;
            pre_slot = vattn_premap(m, pre_len);
        }
        for (uint32_t l = 0; l < cfg.num_layers; l++) {          // what the attention wrapper does before each layer
            if (vattn_wait_layer(m, l) != 0) { printf("wait_layer failed\n"); return 5; }
            if (vattn_layers_ready(m) <= l) { printf("layer %u not ready after wait\n", l); return 6; }
        }
        for (int s = 0; s < 16; s++) {
            if (!lens[s]) continue;
            if (lens[s] >= target[s] || lens[s] >= 4096) {
                if (s & 1) vattn_free_batch_idx_on_stream(m, s, (void*)0x10); else vattn_free_batch_idx(m, s);
                lens[s] = 0;
            }
            else lens[s]++;
        }
        if (it % 97 == 0) vattn_set_deferred_reclamation(m, (it / 97) & 1);
    }
    vattn_stats st;
    vattn_get_stats(m, &st);
    vattn_cleanup(m);
    uint64_t c[12];
    vattn_fake_counters(c);
    printf("layered batches %llu fence waits %llu quiesce %llu\n", (unsigned long long)st.layered_batches, (unsigned long long)st.fence_waits,
           (unsigned long long)st.quiesce_calls);
    if (st.layered_batches == 0) return 7;
    printf("maps %llu unmaps %llu async batches %llu flushes %llu violations %llu mapped-after-cleanup %llu\n", (unsigned long long)st.map_calls,
           (unsigned long long)st.unmap_calls, (unsigned long long)st.async_batches, (unsigned long long)st.tlb_flushes,
           (unsigned long long)c[0], (unsigned long long)c[7]);
    vattn_destroy(m);
    return (c[0] == 0 && c[7] == 0) ? 0 : 4;
}
