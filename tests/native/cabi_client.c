/* A plain C99 consumer of the drop-in boundary (include/vattn.h): what a cgo / JNI / ctypes binding would do, without any
 * C++ or Python in between.  Drives the page manager on the host-only fake backend (tests/native/fake_backend.cpp) through
 * the reference's call sequence init -> reserve -> alloc slot -> step_async -> free slot -> cleanup and prints the
 * observable state; tests/test_cabi_exports.py builds it with gcc and checks the output.
 *   gcc -std=c99 -Iinclude tests/native/cabi_client.c -Lvattention_amd -lvattn_amd -Ltests/native -lvattn_fake_backend */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "vattn.h"
#include "vattn_kernels.h"

extern const vattn_backend_ops* vattn_fake_backend_ops(void);
extern void vattn_fake_reset(uint64_t min_gran, uint64_t rec_gran);

int main(void) {
    vattn_config cfg;
    vattn_t* m = NULL;
    uint64_t lens[4] = {0, 0, 0, 0};
    uint64_t counts[4];
    vattn_layout lay;
    vattn_stats st;
    int rc, slot;
    int64_t pages;

    memset(&cfg, 0, sizeof cfg);
    cfg.num_layers = 2; cfg.num_kv_heads = 2; cfg.head_size = 128; cfg.max_batch_size = 4;
    cfg.max_context_length = 4096; cfg.itemsize = 2; cfg.device = 0; cfg.page_size = 65536; cfg.megacache = 0;
    cfg.flags = VATTN_FLAG_NO_MAPPER_THREAD;
    vattn_fake_reset(4096, 2u << 20);
    rc = vattn_create(&cfg, vattn_fake_backend_ops(), &m);
    if (rc != VATTN_OK) { printf("create failed %d\n", rc); return 1; }
    if (vattn_get_layout(m, &lay) != VATTN_OK) return 1;
    printf("tensors %d ndim %u\n", vattn_num_tensors(m), lay.ndim);
    pages = vattn_reserve_physical_pages(m, 64ull * 65536);
    printf("pool %lld\n", (long long)pages);
    slot = vattn_alloc_new_batch_idx(m, 300);
    printf("slot %d\n", slot);
    lens[slot] = 300;
    rc = vattn_step_async(m, lens, 4);
    rc |= vattn_wait(m);
    printf("step_async %d free_kvblocks %llu\n", rc, (unsigned long long)vattn_num_free_kvblocks(m));
    vattn_get_counts(m, counts);
    vattn_get_stats(m, &st);
    printf("pool_pages %llu mapped_groups %llu needed_groups %llu active_slots %llu map_calls %llu\n", (unsigned long long)counts[0],
           (unsigned long long)counts[1], (unsigned long long)counts[2], (unsigned long long)counts[3], (unsigned long long)st.map_calls);
    lens[0] = 1; lens[1] = 2;                                  /* wrong length argument: explicit error, not a crash */
    rc = vattn_step_async(m, lens, 3);
    printf("bad_len %d err '%s'\n", rc, vattn_last_error(m));
    rc = vattn_free_batch_idx(m, slot);
    printf("free %d\n", rc);
    rc = vattn_cleanup(m);
    printf("cleanup %d\n", rc);
    vattn_destroy(m);
    /* the kernel half of the boundary: argument validation works without a GPU */
    {
        vattn_attn_params p;
        memset(&p, 0, sizeof p);
        rc = vattn_flash_attn_with_kvcache(&p, NULL);
        printf("null_params %d err '%s'\n", rc, vattn_kernels_last_error());
        printf("workspace_bytes %zu sizeof_params %zu\n", vattn_attn_workspace_bytes(&p), sizeof p);
    }
    return 0;
}
