// Fused prefill || decode for hybrid batches — the CDNA4 analogue of the reference's POD-Attention kernel
// (/root/reference/pod_attn/pod_attn/fused_fwd_kernel.h: one CUDA launch whose CTAs are typed prefill / decode by an SM-aware
// counter, so that both operation types are resident on every SM; call site
// /root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_pod_wrapper.py:121-203).
//
// ONE launch of 2 x (number of CUs) PERSISTENT 256-thread workgroups, two per CU (256 registers per lane, 64 KiB of LDS each).
// A workgroup reads its CU's identity (HW_ID / XCC_ID hardware registers) and takes a ticket from that CU's arrival counter:
// the first workgroup to land on a CU prefers PREFILL work, the second prefers DECODE work — every CU then holds one matrix-bound
// wave per SIMD beside one HBM-bound wave per SIMD, which is the co-location two HIP streams cannot promise (a kernel's
// workgroups fill whole CUs first).  Work is handed out by two device-side queues (atomic counters): prefill items are the
// 128-row query blocks (heaviest first), decode items the (sequence, kv head, head block, KV split) tuples; a workgroup whose
// preferred queue has run dry serves the other one, so the launch ends when both are empty, whatever the mix.
// The bodies are the product kernels' own device functions (prefill_body.h: 4 waves x 32 rows; decode_body.h), so the arithmetic
// — and therefore parity — is that of the stand-alone launches.  The split-KV merge of the decode part happens in the same launch:
// the workgroup that completes the last split of a (sequence, kv head, head block) merges its partials (release/acquire through
// __threadfence and the group's counter).
// Control words live at the head of the caller's workspace; they must be zero before the FIRST launch and the kernel leaves
// them zero (the last workgroup to leave resets them), so back-to-back launches on one stream need no memset.
// ROUND 4 — CLOSED: the fused launch is LAB-ONLY (tools/lab/csrc/hybrid_lab.hip, built into tools/lab/libvattn_lab.so only; round 6: this
// product source no longer carries its kernel).  Three rounds of measurements (fused 0.26-0.75x,
// CU-masked streams 0.42-0.93x, two plain streams 0.88-1.19x of the serial order) say that on MI355X each of the two
// stand-alone kernels already owns what bounds it — HBM for decode, board power for prefill — so co-residency creates no capacity; and the
// last candidate, time-slicing the two phases at workgroup granularity inside one persistent launch, can only remove one launch boundary
// (~2 us) and the decode phase's ramp per layer from launches of 0.5-2 ms: < 1 %, against the >= 5 % the row was asked to show.  The
// PRODUCT library therefore implements the C entry point (the reference's POD call site binds it) as what measures best: the plan-chosen
// prefill launch, then the device-planned decode launch, back to back on the caller's stream.
#include "attn_common.h"

namespace vattn_k {

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t hybrid_workspace_bytes(const vattn_attn_params* pp, const vattn_attn_params* pd) {
    return al256(prefill_workspace_bytes(pp)) + al256(decode_workspace_bytes(pd)) + 256;
}
int launch_hybrid(const vattn_attn_params* pp, const vattn_attn_params* pd, void* ws, hipStream_t st) {
    if (pp->dtype != pd->dtype) return fail(VATTN_K_ERR_INVALID, "prefill and decode parts must have the same dtype");
    if (pp->seqlen_q < 2 || pd->seqlen_q != 1) return fail(VATTN_K_ERR_INVALID, "hybrid launch: first part must be a prefill (seqlen_q > 1), second a decode (seqlen_q == 1)");
    if (pp->k_new) return fail(VATTN_K_ERR_UNSUPPORTED, "hybrid launch: append the prefill chunk's keys/values with cache_flat first");
    if (!ws) return fail(VATTN_K_ERR_INVALID, "hybrid launch needs its workspace (vattn_hybrid_workspace_bytes)");
    vattn_attn_params a = *pp, b = *pd;
    a.workspace = ws;
    b.workspace = (char*)ws + al256(prefill_workspace_bytes(pp));
    int rc = launch_prefill_form(&a, st);
    if (rc) return rc;
    return launch_decode_form(&b, st);
}


}  // namespace vattn_k
