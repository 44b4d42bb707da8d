/*
 * vattn_kernels.h — C ABI of the gfx950 attention / KV-append kernels in libvattn_amd.so.
 *
 * These entry points are what the reference's FFI for this path binds:
 *   - vattn_flash_attn_with_kvcache  replaces flash_attn_cuda.fwd_kvcache
 *       (call sites /root/reference/sarathi-lean/sarathi/model_executor/attention/
 *        vattention_flashattention_wrapper.py:159-166 (prefill) and :194-205 (decode);
 *        semantics /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291,
 *        argument rules /root/reference/pod_attn/pod_attn/flash_api.cpp:1291-1578;
 *        the parameter block is the plain-C analogue of Flash_fwd_params,
 *        /root/reference/pod_attn/pod_attn/flash.h:24-154)
 *   - vattn_cache_flat               replaces sarathi.cache_ops.cache_flat
 *       (/root/reference/sarathi-lean/csrc/cache_kernels.cu:482-570, cache.cpp:40-46)
 *
 * All pointers are DEVICE pointers; strides are in ELEMENTS; `stream` is a hipStream_t.
 * Every function returns 0 on success, a negative VATTN_K_* code otherwise, and never touches
 * key/value rows at or beyond a sequence's visible length (those virtual pages may be unmapped).
 */
#ifndef VATTN_KERNELS_H_
#define VATTN_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VATTN_K_OK 0
#define VATTN_K_ERR_UNSUPPORTED (-10)   /* dtype / head_dim / layout not supported */
#define VATTN_K_ERR_INVALID (-11)       /* argument rule violated (message via vattn_kernels_last_error) */
#define VATTN_K_ERR_LAUNCH (-12)        /* HIP launch failure */

#define VATTN_DTYPE_F16 0
#define VATTN_DTYPE_BF16 1

#define VATTN_KERNELS_ABI 5u            /* bumped whenever vattn_attn_params changes */

typedef struct vattn_attn_params {
    /* sizeof(vattn_attn_params) and VATTN_KERNELS_ABI of the header the CALLER was built against.  The block grows between releases
     * (the plan fields below are extensions) and the kernels branch on pointers inside it: a caller built against an older header, or
     * one that does not zero the block, would have the library read past its object.  Every entry point that takes the block checks
     * both words first and refuses a mismatch (VATTN_K_ERR_INVALID).  ZERO the block (memset) before filling it in. */
    uint32_t struct_size;
    uint32_t abi_version;
    /* q / out: [b, seqlen_q, h, d] */
    const void* q;
    void* out;
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    /* caches: [batch_cache, seqlen_k, h_k, d]; last dim contiguous, other strides arbitrary
     * (the decode call passes a [:, :max_cache_len] strided view) */
    void* k_cache;
    void* v_cache;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    /* optional new keys/values [b, seqlen_knew, h_k, d], appended at row cache_seqlens[b] */
    const void* k_new;
    const void* v_new;
    int64_t knew_batch_stride, knew_row_stride, knew_head_stride;
    int64_t vnew_batch_stride, vnew_row_stride, vnew_head_stride;
    const int32_t* cache_seqlens;     /* int32[b] on device, or NULL: every sequence uses seqlen_k      */
    const int32_t* cache_batch_idx;   /* int32[b] on device, or NULL: identity                          */
    float* softmax_lse;               /* optional float[b, h, seqlen_q] (natural log), or NULL          */
    /* split-KV workspace; sized by vattn_attn_workspace_bytes, may be NULL if that returns 0 */
    void* workspace;
    /* batched prefill of several sequences with DIFFERENT chunk lengths (MI355X extension; both NULL otherwise): q / out are
     * then [total_tokens, h, d] (q_batch_stride / o_batch_stride ignored), batch entry i owns rows
     * [q_start[i], q_start[i] + q_lens[i]) and seqlen_q is the maximum of q_lens (it sizes the grid) */
    const int32_t* q_start;           /* int32[b] on device */
    const int32_t* q_lens;            /* int32[b] on device */
    int32_t b, seqlen_q, seqlen_k, seqlen_knew, h, h_k, d;
    int32_t is_causal;                /* bottom-right aligned; ignored when seqlen_q == 1               */
    int32_t dtype;                    /* VATTN_DTYPE_*                                                  */
    int32_t num_splits;               /* 0 = heuristic                                                  */
    float softmax_scale;
    int32_t variant;                  /* 0 = default; debug variants select alternative operand paths   */
    int32_t max_seqlen_k_hint;        /* host-side upper bound of cache_seqlens[b] + seqlen_knew, or 0: unknown (seqlen_k is
                                         only the cache tensor's row count); used to size the prefill KV split            */
    /* Fused rotary position embedding (MI355X extension): when rotary_cos_sin != NULL the launch rotates q — and
     * k_new before it is appended / attended — in registers, NeoX style, with the arithmetic of the reference's stand-alone
     * kernel (/root/reference/sarathi-lean/csrc/pos_encoding_kernels.cu:9-77, every product and the sum rounded to the I/O
     * dtype).  Layout = the reference's cos_sin_cache: row `pos` holds cos[0 .. rotary_dim/2) then sin[0 .. rotary_dim/2), I/O
     * dtype, rows rotary_row_stride elements apart.  Token i of batch entry b sits at position cache_seqlens[b] + i (new keys)
     * resp. (visible keys - seqlen_q) + i (queries), i.e. the call-site convention of the wrapper.  rotary_dim must equal d. */
    const void* rotary_cos_sin;
    int64_t rotary_row_stride;
    int32_t rotary_dim;
    int32_t rotary_reserved;
    /* Length-balanced split-KV of a RAGGED decode batch (MI355X extension; all three zero otherwise).  The reference's heuristic
     * (flash_api.cpp:258-323) gives every sequence of a batch the same number of splits, so a launch lasts as long as the longest
     * sequence's split (a 30 k-token sequence beside 4 k-token ones: 3 x the average).  A caller that knows the lengths on the host
     * (the attention wrapper does) asks vattn_decode_plan for work items of near-equal length — sequence b is cut into
     * ceil(tiles_b / T) pieces of T 32-key tiles — copies the two tables to the device once per iteration and passes them with every
     * layer's call.  decode form only; results equal the uniform split's up to the order of the fp32 merge. */
    const struct vattn_decode_item* split_items;   /* device: num_split_items entries                         */
    const int32_t* split_seq;                      /* device: int32[2 * b] = (first item, item count) per sequence */
    int32_t num_split_items;
    int32_t split_reserved;
    /* Work list of an underfilled / unbalanced PREFILL launch (MI355X extension; all zero otherwise; d = 128).  A causal prompt on a
     * tensor-parallel shard has few heads and query blocks whose key walk grows linearly: the launch lasts as long as its longest
     * block, and cutting EVERY block's key range in n shares multiplies the fp32 partial traffic by n.  vattn_prefill_plan lists the
     * (entry, head, 256-row query block, key-tile range) pieces longest first and cuts only the blocks longer than the per-CU average:
     * short blocks write their output directly, long ones write partials that vattn merges (one wave per row) in a second launch.  The
     * caller copies the tables to the device — once per iteration, the plan depends on the lengths only — and passes them with every
     * layer's call.  Results equal the default launch's up to the order of the fp32 merge. */
    const struct vattn_prefill_item* pf_items;     /* device: num_pf_items pieces, longest first                                */
    const struct vattn_prefill_item* pf_blocks;    /* device: num_pf_blocks SPLIT query blocks (what the merge pass walks)      */
    int32_t num_pf_items, num_pf_blocks;
    int32_t pf_part_rows;                          /* fp32 partial rows of all split blocks (sizes the workspace)               */
    /* PERSISTENT form of the work list (round 5; both zero: one workgroup per piece, in list order).  vattn_prefill_plan_wg also
     * ASSIGNS the pieces: pf_items is grouped by workgroup, workgroup w owns pieces [pf_wg_first[w], pf_wg_first[w + 1]) and walks them
     * in that order without stopping its K / V tile stream between them (csrc/prefill64p_kernels.hip: the next piece's Q block and
     * first tiles are fetched under the current piece's last tiles; one workgroup per CU).  pf_wg_first has pf_num_wg + 1 entries.
     * pf_num_wg > 0 with pf_wg_first == NULL: DRAWN queues — pf_items stays in longest-first order, workgroup w starts with piece w and
     * draws further pieces of its XCD's sub-sequence from a library-owned counter, which balances what a static assignment cannot (the
     * workgroups differ in speed by a few per cent).  What vattn_prefill_plan_wg(wg_first_out = NULL) prepares. */
    int32_t pf_num_wg;
    const int32_t* pf_wg_first;                    /* device: int32[pf_num_wg + 1]                                              */
} vattn_attn_params;

typedef struct vattn_prefill_item {
    int32_t b, h, qb;     /* batch entry, query head, 256-row query block                                                    */
    int32_t tile_begin;   /* first 64-key tile of the piece (pf_blocks: unused)                                              */
    int32_t tile_end;     /* one past its last tile; INT32_MAX for the block's last share: "to the last tile the block sees" —
                             the kernel clamps every range to what the DEVICE-side lengths give, so a list built from stale
                             host lengths costs balance, never keys                                                          */
    int32_t nshares;      /* pieces the query block was cut into; 1 = this piece writes the output rows itself               */
    int32_t part_row;     /* nshares > 1: first of this piece's 256 partial rows (pf_blocks: of the block's share 0; share s
                             sits 256 * s rows further)                                                                      */
    int32_t reserved;     /* 1 on the block's last share                                                                     */
} vattn_prefill_item;

typedef struct vattn_decode_item {
    int32_t b;            /* batch entry                                            */
    int32_t tile_begin;   /* first 32-key tile of the piece                         */
    int32_t tile_end;     /* one past its last tile; INT32_MAX on the sequence's last piece (open-ended, see vattn_prefill_item) */
    int32_t index_in_seq; /* 0 .. count-1 within the sequence                       */
} vattn_decode_item;

/* Bytes of split-KV workspace the call will need: the decode form's partials, or the prefill form's when its grid
 * would underfill the chip and the key range is split across workgroups (0 otherwise). */
size_t vattn_attn_workspace_bytes(const vattn_attn_params* p);

/* Host-side planner of the length-balanced decode split (see vattn_attn_params.split_items).  `p` describes the call (b, h, h_k, d,
 * seqlen_knew, variant; pointers are not read), cache_seqlens_host[b] are the values the device array will hold.  Writes at most
 * `cap` items and 2 * b ints of (first item, count); returns the number of items, 0 when the uniform split is at least as good
 * (equal lengths, one sequence, batches whose uniform split is already balanced) or the tables would not fit, < 0 on bad arguments.
 * p->num_splits = -T forces pieces of T tiles (tests, A/B measurements); the call itself is then made with num_splits = 0.
 * Pure host arithmetic (no device access): usable, and tested, without a GPU. */
int32_t vattn_decode_plan(const vattn_attn_params* p, const int32_t* cache_seqlens_host, vattn_decode_item* items_out, int32_t cap,
                          int32_t* seq_out);

/* Host-side planner of the prefill work list (see vattn_attn_params.pf_items).  `p` describes the call (b, seqlen_q, h, h_k, d,
 * is_causal; pointers are not read); q_lens_host[b] are the chunk lengths (NULL: every entry has seqlen_q rows), k_lens_host[b] the
 * visible keys of each entry (cache length + new tokens).  Writes at most cap_items / cap_blocks entries and counts_out[3] =
 * {items, split blocks, partial rows}; returns the number of items, 0 when the default launch is at least as good (grids that fill
 * the chip with balanced work, short key walks, head dimensions other than 128) or a table would not fit, < 0 on bad arguments.
 * p->num_splits = -T forces pieces of at most T tiles (tests, A/B measurements).  Pure host arithmetic. */
int32_t vattn_prefill_plan(const vattn_attn_params* p, const int32_t* q_lens_host, const int32_t* k_lens_host, vattn_prefill_item* items_out,
                           int32_t cap_items, vattn_prefill_item* blocks_out, int32_t cap_blocks, int32_t* counts_out);

/* The same plan for PERSISTENT workgroups (vattn_attn_params.pf_num_wg / pf_wg_first): pieces are priced with the chained overhead
 * (a piece that follows another in a workgroup's queue pays its epilogue and a fragment of a tile, not a cold prologue), so finer cuts
 * pay off, and are assigned to at most `max_wg` workgroups (<= 0: one per CU, 256) longest first, each to the least loaded one among
 * the workgroups of its kv head's XCD class (workgroup w runs on XCD w % 8; class = kv head modulo the classes that divide 8, so that an
 * XCD's L2 keeps seeing one kv head).  items_out comes back GROUPED by workgroup; wg_first_out receives num_wg + 1 offsets — THE
 * CALLER PROVIDES ROOM FOR (max_wg > 0 ? max_wg : 256) + 1 = at most 257 int32 VALUES (the function has no capacity argument for it);
 * counts_out[4] = {items, split blocks, partial rows, num_wg}.  wg_first_out == NULL: no assignment (drawn queues, see pf_num_wg):
 * items_out stays longest first, only num_wg is chosen.  Returns the number of items (0: default launch, as above). */
int32_t vattn_prefill_plan_wg(const vattn_attn_params* p, const int32_t* q_lens_host, const int32_t* k_lens_host, vattn_prefill_item* items_out,
                              int32_t cap_items, vattn_prefill_item* blocks_out, int32_t cap_blocks, int32_t* wg_first_out, int32_t max_wg,
                              int32_t* counts_out);

/* What vattn_flash_attn_with_kvcache will launch for `p` — pure host arithmetic on the shapes (pointers are only tested for NULL), so a
 * test can pin the launch plans without a GPU and without a stopwatch (tests/test_plan_table.py; the reference's equivalents are the
 * launch heuristics of flash_api.cpp:258-323 and flash_fwd_launch_template.h:100-162).  Returns 0, or VATTN_K_ERR_INVALID. */
typedef struct vattn_plan_desc {
    int32_t form;          /* 0 = prefill (seqlen_q > 1), 1 = decode                                                              */
    int32_t path;          /* prefill: 0 = grid order, 1 = work list (pf_items).  decode: 0 = uniform split of every sequence (grid
                              heuristics), 1 = host item plan (split_items), 2 = device-planned stream decomposition               */
    int32_t tiling;        /* prefill: 1 = 8 waves x 32 rows, 4 = 4 waves x 32 rows, 7 = prefill64 (4 waves x 64 rows).
                              decode: 16-head blocks per workgroup (1 or 2)                                                        */
    int32_t nsplit;        /* key-range shares per work item of the grid paths (1 = none); 0 on the list / item / stream paths      */
    int32_t workgroups;    /* workgroups of the main launch that hold work (grid padding excluded)                                 */
    int32_t merge_launch;  /* 1 = a second launch merges fp32 partials                                                             */
    int64_t workspace_bytes;
} vattn_plan_desc;
int vattn_attn_plan_describe(const vattn_attn_params* p, vattn_plan_desc* out);

/* flash_attn_with_kvcache: appends k_new/v_new (if given) and attends; prefill form (seqlen_q > 1,
 * causal chunk against the growing cache) and decode form (seqlen_q == 1, split-KV + combine). */
int vattn_flash_attn_with_kvcache(const vattn_attn_params* p, void* stream);

/* Fused prefill || decode for a hybrid batch (replaces the reference's POD-Attention entry point
 * /root/reference/pod_attn/pod_attn/flash_attn_interface.py true_fused_attn_with_kvcache, call site
 * /root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_pod_wrapper.py:121-203): ONE launch of
 * persistent workgroups, two per CU, typed prefill / decode by a per-CU arrival counter, fed from two device-side work queues;
 * the decode part's split-KV merge happens in the same launch.  `prefill` is a prefill-form parameter block (seqlen_q > 1 or the
 * batched-chunk form, k_new == NULL: append with vattn_cache_flat first), `decode` a decode-form block (seqlen_q == 1, optional
 * one-row append); both d == 128 and the same dtype; their `workspace` fields are ignored.  `workspace` is
 * vattn_hybrid_workspace_bytes() bytes of device memory that must be ZERO before the first launch; the launch leaves its control
 * words and merge counters zero (they sit in a fixed-size region ahead of the split partials, whatever the batch), so it can be
 * reused by the next launch on the same stream — of any batch size it is large enough for — without a memset.  Results are those of the two
 * stand-alone launches (same device functions). */
size_t vattn_hybrid_workspace_bytes(const vattn_attn_params* prefill, const vattn_attn_params* decode);
int vattn_hybrid_attn(const vattn_attn_params* prefill, const vattn_attn_params* decode, void* workspace, void* stream);

/* cache_flat: k_cache[t*k_cache_stride + i] = key[t*key_stride + i], same for value,
 * t < num_tokens, i < num_heads*head_size (cache_kernels.cu:483-520). */
int vattn_cache_flat(const void* key, const void* value, void* k_cache, void* v_cache,
                     int64_t num_tokens, int32_t num_heads, int32_t head_size,
                     int64_t key_stride, int64_t value_stride, int64_t k_cache_stride, int64_t v_cache_stride,
                     int32_t itemsize, void* stream);

/* cache_flat with the rotary embedding of the KEY rows fused in: k_cache[t] = rope(key[t], position pos0 + t),
 * v_cache[t] = value[t]; one pass over the new K/V instead of the reference's rotary kernel + cache_flat.  2-byte dtypes, NeoX
 * style, rotary_dim == head_size.  dtype = VATTN_DTYPE_*. */
int vattn_cache_flat_rope(const void* key, const void* value, void* k_cache, void* v_cache, int64_t num_tokens,
                          int32_t num_heads, int32_t head_size, int64_t key_stride, int64_t value_stride, int64_t k_cache_stride,
                          int64_t v_cache_stride, int32_t dtype, const void* cos_sin, int64_t cos_sin_row_stride, int64_t pos0,
                          void* stream);

/* The reference's stand-alone rotary kernel (pos_encoding_kernels.cu:39-77, rotary_embedding): in place on query [T, Hq*hs]
 * and key [T, Hkv*hs], positions int64[T].  Provided for the UNFUSED path (what the fused launches are measured against). */
int vattn_rotary_embedding(const int64_t* positions, void* query, void* key, int64_t num_tokens, int32_t num_q_heads,
                           int32_t num_kv_heads, int32_t head_size, int64_t query_stride, int64_t key_stride, int32_t dtype,
                           const void* cos_sin, int64_t cos_sin_row_stride, int32_t rot_dim, int32_t is_neox, void* stream);

/* Device-side self-tests of the hardware layout assumptions (MFMA fragment maps, LDS transpose read, LDS-DMA lane
 * mapping); 0 = all assumptions hold.  detail_out[0..5] are per-assumption failure flags, detail_out[6] reports what
 * an out-of-range LDS-DMA lane does to its LDS bytes (0 zeros, 1 untouched).  Used by tests/test_gpu_hw_and_vmm.py. */
int vattn_selftest_layouts(void* stream, int32_t* detail_out /* int32[8] host buffer */);

/* Times `iters` launches of the kernel family last used by vattn_flash_attn_with_kvcache with HIP
 * events on `stream`; returns average milliseconds per call (negative on error). */
float vattn_time_attn(const vattn_attn_params* p, void* stream, int32_t warmup, int32_t iters);

const char* vattn_kernels_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VATTN_KERNELS_H_ */
