/*
 * vattn.h — C ABI of the MI355X-native vAttention page manager (libvattn_amd.so).
 *
 * Drop-in boundary for the reference's `vattention` Python module
 * (/root/reference/vattention/vattention.cu:614-637, apis.h:1-63).  Each entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no torch types, no exceptions:
 * every call returns a VATTN_* code (or a value whose negative range is the code) and
 * vattn_last_error() holds the message the Python layer turns into the reference's exception.
 *
 * Execution model (differs from the reference by design, observable bookkeeping is identical):
 *   - integer bookkeeping (slots, mapped page counts, pool) is updated synchronously inside the
 *     call, so num_free_kvblocks / alloc_new_batch_idx never race with background work;
 *   - the driver calls a step needs NOW are executed before the call returns;
 *   - look-ahead mapping / reclamation planned by step_async is executed by ONE mapper thread
 *     while the GPU runs the forward pass; the next step()/step_async()/cleanup() joins it first
 *     (HIP has no stream-ordered VMM call: hipMemMapArrayAsync returns hipErrorNotSupported,
 *     /opt/rocm/include/hip/hip_runtime_api.h:9409-9420).
 */
#ifndef VATTN_H_
#define VATTN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VATTN_OK 0
#define VATTN_ERR_INVALID (-1)     /* bad argument / bad state                                  */
#define VATTN_ERR_OOM (-2)         /* "OOM on demand" — vattention.cu:295                        */
#define VATTN_ERR_DRIVER (-3)      /* a HIP VMM call failed (reference: exit(1), cudaInternal.h:1-13) */
#define VATTN_ERR_POOL_EMPTY (-4)  /* "page pool is empty" — mux.h:2-3                           */

/* vattn_config.flags */
#define VATTN_FLAG_EAGER_CREATE 1u      /* create every physical handle inside reserve (reference behaviour) */
#define VATTN_FLAG_NO_ACCESS_MERGE 2u   /* one set-access call per page instead of one per contiguous run   */
#define VATTN_FLAG_NO_MAPPER_THREAD 4u  /* run "background" work inline (deterministic tests)               */
/* step_async maps a new prompt's pages LAYER-ORDERED: layers [0, sync_layers) before it returns, the remaining layers on the
 * mapper thread, which publishes a per-layer ready count; the caller must then call vattn_wait_layer(l) before it launches
 * layer l's kernels (the attention wrapper does).  Hides (L - sync_layers)/L of new-prompt mapping under the layers already
 * running.  Opt-in: an engine that does not call vattn_wait_layer must not set it.  Ignored with megacache (one page covers
 * every layer) and for batches that unmap. */
#define VATTN_FLAG_LAYERED_ASYNC 8u
#define VATTN_FLAG_NO_VMM_SELFCHECK 16u /* skip the remap/TLB self-check the HIP backend runs once per device at create  */

typedef struct vattn_config {
    uint32_t num_layers;          /* init_kvcache(num_layers, ...)   apis.h:3-13 */
    uint32_t num_kv_heads;
    uint32_t head_size;
    uint32_t max_batch_size;
    uint64_t max_context_length;
    uint32_t itemsize;            /* dtype.itemsize (vattention.cu:122) */
    int32_t device;
    uint64_t page_size;           /* bytes; any multiple of the HIP VMM minimum granularity */
    uint32_t megacache;
    uint32_t flags;
} vattn_config;

/* Pluggable physical backend.  NULL selects the HIP VMM backend (the product path).  A custom
 * table exists so the bookkeeping core can be driven on a CPU-only box by tests (tests/native/)
 * and under ThreadSanitizer; it is never selected implicitly. All functions return 0 on success. */
typedef struct vattn_backend_ops {
    void* ctx;
    int (*granularity)(void* ctx, uint64_t* min_gran, uint64_t* rec_gran);
    int (*reserve_va)(void* ctx, uint64_t bytes, uint64_t align, uint64_t* base_out);
    int (*free_va)(void* ctx, uint64_t base, uint64_t bytes);
    int (*create)(void* ctx, uint64_t bytes, uint64_t* handle_out);
    int (*release)(void* ctx, uint64_t handle);
    int (*map)(void* ctx, uint64_t va, uint64_t bytes, uint64_t handle);
    int (*set_access)(void* ctx, uint64_t va, uint64_t bytes);
    int (*unmap)(void* ctx, uint64_t va, uint64_t bytes);
    int (*thread_init)(void* ctx);          /* called once on the mapper thread (may be NULL) */
    /* Make every earlier unmap visible to the GPU (TLB invalidation).  On ROCm 7.2 / gfx950 a kernel keeps
     * using the OLD translation of a virtual page after hipMemUnmap (+ hipMemMap of another handle at the
     * same address) until the driver next services an ordinary allocation (tools/remap_probe*.cpp,
     * profiles/r01_remap_probe3_raw.txt, r01_remap_probe4_raw.txt); the manager therefore calls this once after every batch that unmapped
     * anything, before the batch is reported complete.  May be NULL (no-op). */
    int (*tlb_flush)(void* ctx);
    /* Wait until every kernel already queued on the device has finished.  Called once per batch before its first
     * unmap: the engine frees a slot right after *launching* the iteration that still reads its pages, and on
     * ROCm 7.2 hipMemUnmap only waits for blocking streams (profiles/r01_vmm_sync_probe_raw.txt) — with compute on
     * a non-blocking stream an unmap could pull a page from under a running kernel (GPU memory fault).  May be NULL. */
    int (*quiesce)(void* ctx);
    /* Free/total device memory in bytes; reserve_physical_pages refuses a pool the device cannot back (handles are created
     * lazily, so without this an over-sized pool would only fail mid-serving).  May be NULL (no check). */
    int (*mem_info)(void* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
    /* Per-slot fences, replacing the device-wide quiesce where the engine cooperates: fence_record(slot, stream) is called by
     * vattn_free_batch_idx_on_stream — it marks the point in `stream` after the last kernel that can read the slot's pages
     * (stream == NULL clears the slot's fence); fence_wait(slot) is called before the first unmap of a freed slot's pages and
     * returns 0 once that point has passed, 1 if the slot has no fence (the manager then falls back to quiesce), < 0 on
     * error.  Both may be NULL. */
    int (*fence_record)(void* ctx, uint32_t slot, void* stream);
    int (*fence_wait)(void* ctx, uint32_t slot);
} vattn_backend_ops;

typedef struct vattn_layout {       /* element-unit description of every returned tensor */
    uint32_t ndim;                  /* 4, or 5 for megacache */
    uint64_t shape[5];              /* [B, max_ctx, kvh, D] or [B, max_ctx, L, kvh, D]  (vattention.cu:142-153) */
    uint64_t stride[5];             /* batch stride = virt_bytes_per_req / itemsize (explicit padding) */
    uint64_t virt_bytes_per_req;    /* ROUND_UP(max_ctx*row_bytes, page)  vattention.cu:57-58 */
    uint64_t virt_bytes_total;
    uint64_t tokens_per_page;       /* vattention.cu:38-47 */
    uint64_t max_pages_per_req;
    uint64_t page_size;
} vattn_layout;

typedef struct vattn_stats {
    uint64_t handles_created, handles_released;
    uint64_t map_calls, access_calls, unmap_calls;
    uint64_t sync_batches, async_batches;
    uint64_t sync_ns, async_ns;           /* wall time inside driver calls, per class */
    uint64_t join_wait_ns;                /* time step()/step_async() spent waiting for the mapper */
    uint64_t create_ns;
    uint64_t pages_mapped_now;            /* currently mapped physical pages */
    uint64_t tlb_flushes, tlb_flush_ns;
    uint64_t quiesce_calls, quiesce_ns;
    uint64_t fence_waits, fence_wait_ns;  /* per-slot fences waited on instead of a device-wide quiesce */
    uint64_t layered_batches, layer_wait_ns;   /* VATTN_FLAG_LAYERED_ASYNC: batches split by layer; time vattn_wait_layer blocked */
    uint64_t rollbacks;                   /* batches whose unexecuted maps were rolled back after a driver failure */
    /* what the synchronous batches' time (sync_ns) was spent on besides map / set-access / unmap calls */
    uint64_t sync_create_ns, sync_creates;    /* handles that had to be created on the critical path */
    uint64_t sync_fence_ns;                   /* waiting for slot fences / device quiesce before an unmap */
    uint64_t sync_tlb_ns;                     /* TLB invalidation after unmaps */
    uint64_t sync_maps, sync_unmaps;          /* driver calls executed on the critical path */
} vattn_stats;

typedef struct vattn_handle vattn_t;

/* init_kvcache (apis.h:3-13; vattention.cu:97-128,142-187): validates the configuration, reserves
 * 2*L (or 2) virtual ranges.  Tensor i: K_l -> l, V_l -> L+l (megacache: K -> 0, V -> 1). */
int vattn_create(const vattn_config* cfg, const vattn_backend_ops* backend_or_null, vattn_t** out);
int vattn_num_tensors(const vattn_t* m);
uint64_t vattn_tensor_base(const vattn_t* m, int i);          /* device VA of tensor i */
int vattn_get_layout(const vattn_t* m, vattn_layout* out);

/* reserve_physical_pages (apis.h:23-25; cudaInternal.h:45-59; utils.h:221-228): returns the pool
 * size in pages (rounded down to a multiple of 2*L), negative VATTN_ERR_* on failure. */
int64_t vattn_reserve_physical_pages(vattn_t* m, uint64_t free_memory);

/* step (apis.h:27-29; vattention.cu:395-409) and step_async (apis.h:31-35; vattention.cu:549-558).
 * n must equal max_batch_size (the reference indexes unchecked). */
int vattn_step(vattn_t* m, const uint64_t* seq_lens, uint32_t n, int eager_reclaim);
int vattn_step_async(vattn_t* m, const uint64_t* seq_lens, uint32_t n);
int vattn_wait(vattn_t* m);                                    /* join outstanding background mapping */

/* alloc_new_batch_idx / free_batch_idx / num_free_kvblocks (apis.h:53-63; vattention.cu:189-217,564-594) */
int vattn_alloc_new_batch_idx(vattn_t* m, uint64_t seqlen);    /* slot, or -1 if none is free */
int vattn_free_batch_idx(vattn_t* m, int slot);
/* free_batch_idx + a fence on `stream` (a hipStream_t; the stream the iteration that last reads the slot was launched on):
 * a later reclaim of the slot's pages waits for that point only instead of synchronising the whole device. */
int vattn_free_batch_idx_on_stream(vattn_t* m, int slot, void* stream);
/* Admission look-ahead (MI355X extension): reserve the slot alloc_new_batch_idx(seqlen) would return and map the pages `seqlen`
 * tokens need on the MAPPER thread, while the current iteration runs.  Returns the slot (-1: none free).  The slot stays
 * inactive until its length is passed to vattn_step / vattn_step_async (which then maps nothing for it on the critical
 * path); alloc_new_batch_idx skips reserved slots; vattn_free_batch_idx / vattn_cancel_premap release the reservation. */
int vattn_premap(vattn_t* m, uint64_t seqlen);
int vattn_cancel_premap(vattn_t* m, int slot);
/* Lazy pool (the default: handles are created on first map or by the idle mapper thread): block until the mapper has created the
 * handles it creates ahead of demand — the WHOLE pool when it has at most 40 000 pages, a window of 4 096 below the pool's frontier
 * otherwise — or `timeout_ms` have passed (< 0: no limit).  Returns the number of handles still to be created ahead of demand (0 =
 * ready; always 0 with VATTN_FLAG_EAGER_CREATE / VATTN_FLAG_NO_MAPPER_THREAD), or a negative VATTN_ERR_* when a creation ahead of
 * demand failed (VATTN_ERR_DRIVER: the pool does not fit the device — where the reference aborts inside reserve) or the mapper is dead.  The reference commits every page inside
 * reserve_physical_pages (/root/reference/vattention/cudaInternal.h:45-59); an engine calls this once after reserve, before it admits
 * requests, so that no launch of the first iterations shares the driver with a burst of hipMemCreate calls (round 5:
 * profiles/r05_cold_pool.md — the kernels are not slowed by concurrent creation, the LAUNCHES are late).  Thread-safe. */
int64_t vattn_wait_pool_ready(vattn_t* m, int64_t timeout_ms);
/* VATTN_FLAG_LAYERED_ASYNC: block until the pages the current step needs are mapped for `layer` (returns at once when no
 * layered batch is pending); VATTN_ERR_* if the mapper failed (vattn_last_error is NOT updated by this call — it may run
 * concurrently with the engine thread's calls; the next vattn_step* / vattn_wait reports the driver's message).  A caller that
 * does not know its layer index waits for num_layers - 1 once per iteration.  vattn_layers_ready: layers mapped so far (num_layers when
 * nothing is pending).  vattn_set_sync_layers: layers mapped before step_async returns (default 2). */
int vattn_wait_layer(vattn_t* m, uint32_t layer);
uint32_t vattn_layers_ready(vattn_t* m);
int vattn_set_sync_layers(vattn_t* m, uint32_t n);
uint64_t vattn_num_free_kvblocks(vattn_t* m);                  /* u64 wrap-around kept (utils.h:177-183) */

/* set_deferred_reclamation / set_verbose / map_common_pages / show_* (apis.h:15-21,37-51) */
int vattn_set_deferred_reclamation(vattn_t* m, int on);
int vattn_set_verbose(vattn_t* m, int on);
int vattn_map_common_pages(vattn_t* m, uint64_t num_tokens);
int vattn_show_kvcache_config(vattn_t* m);
int vattn_show_allocator_state(vattn_t* m);

/* cleanup (apis.h:41-43; vattention.cu:601-609; cudaInternal.h:84-94): joins the mapper, unmaps
 * everything, frees the virtual ranges, releases every physical handle. */
int vattn_cleanup(vattn_t* m);
void vattn_destroy(vattn_t* m);

/* Introspection for bit-exact tests.  out = [B, pool_size, pagemap_rows, mapped[B], lens[B],
 * pool page-ids bottom..top]; returns words written, or -(words needed). */
int64_t vattn_state_dump(vattn_t* m, uint64_t* out, uint64_t cap);
/* page map rows (slot, byte offset, layer, k page-id, v page-id) in key order; returns rows. */
int64_t vattn_pagemap_dump(vattn_t* m, uint64_t* out, uint64_t cap_rows);
int vattn_get_stats(vattn_t* m, vattn_stats* out);
/* O(max_batch_size) summary: out = [pool pages, sum of mapped page-groups, sum of page-groups needed by the
 * current lengths, active slots]. */
int vattn_get_counts(vattn_t* m, uint64_t out[4]);
const char* vattn_last_error(const vattn_t* m);
/* Remap / TLB self-check of the HIP VMM backend on `device` (what vattn_create runs once per device): maps page A at a
 * virtual address, lets a kernel read it, unmaps, maps page B at the same address, applies the backend's TLB-invalidation
 * policy and lets a kernel read again.  0 = the kernel sees B; 1 = STALE translation survived (vattn_create refuses to
 * start: silent cross-request KV corruption); < 0 = driver error.  detail (may be NULL) = {value read before, value read
 * after WITHOUT the flush, value read after the flush}. */
int vattn_vmm_selfcheck(int device, uint32_t detail[3]);
/* HIP VMM granularity probe for a device: 0 on success. */
int vattn_hip_granularity(int device, uint64_t* min_gran, uint64_t* rec_gran);
/* HIP runtime / driver versions (hipRuntimeGetVersion / hipDriverGetVersion) of the process.  The TLB-invalidation step of the unmap
 * policy rests on observed driver behaviour, not on an API contract (csrc/hip_backend.cpp, profiles/r06_tlb_flush_probe.txt: nothing the
 * VMM API itself offers carries the invalidation on ROCm 7.2); it is therefore PROVEN on the device by vattn_vmm_selfcheck at every
 * vattn_create, whatever the version, and a failed proof names these versions in vattn_last_error.  0 on success. */
int vattn_hip_versions(int* runtime_version, int* driver_version);

#ifdef __cplusplus
}
#endif
#endif /* VATTN_H_ */
