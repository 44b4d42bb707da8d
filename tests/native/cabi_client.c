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
    printf("pool_ready %lld\n", (long long)vattn_wait_pool_ready(m, 1000));      /* 0: nothing left to create ahead of demand (inline mode) */
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
        p.struct_size = (uint32_t)sizeof p; p.abi_version = VATTN_KERNELS_ABI;
        rc = vattn_flash_attn_with_kvcache(&p, NULL);
        printf("null_params %d err '%s'\n", rc, vattn_kernels_last_error());
        printf("workspace_bytes %zu sizeof_params %zu\n", vattn_attn_workspace_bytes(&p), sizeof p);
    }
    /* the host-side planners (pure arithmetic): a ragged decode batch cut into near-equal pieces, a tensor-parallel shard's prompt
     * listed longest piece first with only its long query blocks cut; a lab-only variant is refused by the product library */
    {
        vattn_attn_params p;
        static vattn_decode_item ditems[4096];
        static vattn_prefill_item pitems[4096], pblocks[512];
        int32_t dlens[64], dseq[128], qlen[1] = {8192}, klen[1] = {8192}, counts[3];
        int32_t n, i, longest = 0;
        memset(&p, 0, sizeof p);
        p.struct_size = (uint32_t)sizeof p; p.abi_version = VATTN_KERNELS_ABI;
        p.b = 64; p.seqlen_q = 1; p.seqlen_k = 32768; p.seqlen_knew = 1; p.h = 8; p.h_k = 1; p.d = 128;
        for (i = 0; i < 64; i++) dlens[i] = 500 + 450 * i;                         /* 500 .. 28 850 tokens */
        n = vattn_decode_plan(&p, dlens, ditems, 4096, dseq);
        for (i = 0; i < n; i++) {      /* (a sequence's last piece is open-ended, INT32_MAX: it runs to the sequence's last tile) */
            const int32_t tiles = (dlens[ditems[i].b] + 1 + 31) / 32;
            const int32_t te = ditems[i].tile_end < tiles ? ditems[i].tile_end : tiles;
            if (te - ditems[i].tile_begin > longest) longest = te - ditems[i].tile_begin;
        }
        printf("decode_plan items %d first_seq_pieces %d last_seq_pieces %d longest_piece_tiles %d\n", n, dseq[1], dseq[127], longest);
        memset(&p, 0, sizeof p);
        p.struct_size = (uint32_t)sizeof p; p.abi_version = VATTN_KERNELS_ABI;
        p.b = 1; p.seqlen_q = 8192; p.h = 8; p.h_k = 1; p.d = 128; p.is_causal = 1;
        n = vattn_prefill_plan(&p, qlen, klen, pitems, 4096, pblocks, 512, counts);
        printf("prefill_plan items %d split_blocks %d partial_rows %d first_piece_tiles %d last_piece_tiles %d\n", n, counts[1], counts[2],
               /* (a block's last share is open-ended: causal whole prompt, query block qb sees 4 (qb + 1) tiles of 64 keys) */
               (pitems[0].tile_end < 4 * (pitems[0].qb + 1) ? pitems[0].tile_end : 4 * (pitems[0].qb + 1)) - pitems[0].tile_begin,
               (pitems[n > 0 ? n - 1 : 0].tile_end < 4 * (pitems[n > 0 ? n - 1 : 0].qb + 1) ? pitems[n > 0 ? n - 1 : 0].tile_end : 4 * (pitems[n > 0 ? n - 1 : 0].qb + 1)) - pitems[n > 0 ? n - 1 : 0].tile_begin);
        {   /* the same prompt for PERSISTENT workgroups (round 5): pieces assigned to at most 64 queues, grouped by queue */
            static int32_t wg_first[257];
            int32_t c4[4];
            int nw, w, ok = 1;
            n = vattn_prefill_plan_wg(&p, qlen, klen, pitems, 4096, pblocks, 512, wg_first, 64, c4);
            nw = c4[3];
            for (w = 0; w < nw; w++) ok &= wg_first[w] < wg_first[w + 1];
            printf("prefill_plan_wg items %d queues %d first %d last %d nonempty %d\n", n, nw, wg_first[0], wg_first[nw], ok);
            /* ... and with drawn queues: nothing assigned, the list stays longest first */
            n = vattn_prefill_plan_wg(&p, qlen, klen, pitems, 4096, pblocks, 512, NULL, 0, c4);
            printf("prefill_plan_drawn items %d queues %d\n", n, c4[3]);
        }
        {   /* what WILL be launched, asked on the host (no GPU): configs[1]'s whole prompt and its batch-16 decode step */
            vattn_attn_params d;
            vattn_plan_desc desc;
            memset(&d, 0, sizeof d);
            d.struct_size = (uint32_t)sizeof d; d.abi_version = VATTN_KERNELS_ABI;
            d.b = 1; d.seqlen_q = 32702; d.seqlen_k = 32768; d.h = 32; d.h_k = 4; d.d = 128; d.is_causal = 1; d.dtype = VATTN_DTYPE_F16;
            rc = vattn_attn_plan_describe(&d, &desc);
            printf("describe_prefill %d form %d path %d tiling %d nsplit %d workgroups %d merge %d\n", rc, desc.form, desc.path, desc.tiling, desc.nsplit,
                   desc.workgroups, desc.merge_launch);
            d.b = 16; d.seqlen_q = 1; d.seqlen_knew = 1;
            rc = vattn_attn_plan_describe(&d, &desc);
            printf("describe_decode %d form %d path %d tiling %d workgroups %d merge %d workspace %lld\n", rc, desc.form, desc.path, desc.tiling,
                   desc.workgroups, desc.merge_launch, (long long)desc.workspace_bytes);
            d.abi_version = VATTN_KERNELS_ABI - 1;     /* a block built against another header is refused, not read */
            printf("describe_other_abi %d\n", vattn_attn_plan_describe(&d, &desc));
        }
        p.q = p.out = p.k_cache = p.v_cache = &p;      /* (never dereferenced: validation stops at the variant) */
        p.seqlen_k = 8192; p.dtype = VATTN_DTYPE_F16;
        p.q_row_stride = p.q_head_stride = p.q_batch_stride = p.k_row_stride = p.k_head_stride = p.k_batch_stride = 8;
        p.v_row_stride = p.v_head_stride = p.v_batch_stride = p.o_row_stride = p.o_head_stride = p.o_batch_stride = 8;
        p.variant = 4 << 8;                            /* a timing ablation of the lab build */
        rc = vattn_flash_attn_with_kvcache(&p, NULL);
        printf("lab_variant %d err '%s'\n", rc, vattn_kernels_last_error());
    }
    return 0;
}
