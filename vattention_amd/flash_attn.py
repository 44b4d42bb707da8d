"""Drop-in for the part of the `flash_attn` package the reference's vAttention wrappers use:
`flash_attn_with_kvcache` and `flash_attn_func`
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:4,159-166,194-205).

Signature and semantics follow /root/reference/pod_attn/pod_attn/flash_attn_interface.py:1146-1291;
the work is done by the gfx950 kernels in libvattn_amd.so through the C ABI
(include/vattn_kernels.h) on torch's current HIP stream.  Arguments this path never uses (paged
block_table, alibi, sliding window, softcap, leftpad) raise NotImplementedError.

Rotary embedding (`rotary_cos` / `rotary_sin` [seqlen_ro, rotary_dim/2], or `_rotary_cos_sin` = the
reference model's own cos_sin_cache [max_position, rotary_dim]) is FUSED into the launch:
q and the new k are rotated in registers, the rotated k is what lands in the cache.  NeoX pairing only
(`rotary_interleaved=False`), rotary_dim == head size, and the arithmetic is that of the reference's
stand-alone kernel (sarathi-lean/csrc/pos_encoding_kernels.cu: products and sum rounded to the I/O dtype),
because that is what this path's model code applies (models/yi.py:172-173), not FlashAttention's fp32 form.

Deliberate difference: the reference raises "If key is supplied, it must have seqlen
<= the seqlen of the KV cache" when the GQA-swapped seqlen_q exceeds the cache view (contexts shorter
than Hq/Hkv tokens) and its wrapper then silently returns unwritten output; this shim raises that
message only when the new keys genuinely do not fit, and otherwise always computes.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Union

import torch

from . import kernels as K

APPEND_ERR = "If key is supplied, it must have seqlen <= the seqlen of the KV cache"
_workspaces = {}
_STREAM_SWITCH = 0      # tuning hook of tools/: 1 + switch allowance of the stream decode plan (0: the library's default)
_rotary_cat = {}      # (id(cos), id(sin), versions, dtype) -> (cos, sin, the [S, rotary_dim] cat(cos, sin) tensor the kernels read)


def _rotary_table(rotary_cos, rotary_sin, _rotary_cos_sin, rotary_interleaved, q):
    if _rotary_cos_sin is None and rotary_cos is None and rotary_sin is None:
        return None
    if _rotary_cos_sin is None:
        if rotary_interleaved:      # (the model-side table `_rotary_cos_sin` is NeoX by construction: models/yi.py is_neox_style=True)
            raise NotImplementedError("fused rotary embedding implements the NeoX pairing only (pass rotary_interleaved=False)")
        if rotary_cos is None or rotary_sin is None:
            raise RuntimeError("rotary_cos and rotary_sin must be given together")
        # the entry holds the source tensors (their ids / addresses cannot be recycled while it lives) and is keyed on their
        # versions (an in-place update of the tables makes a new entry) and on the query dtype
        key = (id(rotary_cos), id(rotary_sin), rotary_cos._version, rotary_sin._version, q.dtype)
        ent = _rotary_cat.get(key)
        if ent is None:
            if len(_rotary_cat) > 8:
                _rotary_cat.clear()
            ent = _rotary_cat[key] = (rotary_cos, rotary_sin, torch.cat((rotary_cos, rotary_sin), dim=-1).to(q.dtype).contiguous())
        _rotary_cos_sin = ent[2]
    t = _rotary_cos_sin
    if t.dtype != q.dtype or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("rotary cos/sin table must be a [positions, rotary_dim] GPU tensor of the query's dtype")
    if t.shape[1] != q.shape[-1]:
        raise NotImplementedError("fused rotary embedding needs rotary_dim == head size")
    return t


def _workspace(nbytes: int, device, stream_ptr=None) -> torch.Tensor:
    # one scratch buffer per (device, stream): split-KV partials of calls on different streams must not share storage
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           stream_ptr if stream_ptr is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def _set_cache(p, k_cache, v_cache):
    p.k_cache, p.v_cache = k_cache.data_ptr(), v_cache.data_ptr()
    p.k_batch_stride, p.k_row_stride, p.k_head_stride = k_cache.stride(0), k_cache.stride(1), k_cache.stride(2)
    p.v_batch_stride, p.v_row_stride, p.v_head_stride = v_cache.stride(0), v_cache.stride(1), v_cache.stride(2)


_capture = None      # a list while hybrid_attn() records the two parameter blocks of a fused prefill || decode launch
_hybrid_ws = {}      # (device, stream) -> zero-initialised workspace of the fused launch (the kernel leaves its control words zero)


def hybrid_attn(prefill_call, decode_call, device, _role_mode: int = 0, _product: bool = False) -> None:
    """LAB ONLY (tools/lab/libvattn_lab.so; the product's vattn_hybrid_attn issues the two launches back to back — what measures best).
    Fused prefill || decode for a hybrid batch (include/vattn_kernels.h, vattn_hybrid_attn; the reference's POD entry point
    pod_attn/flash_attn_interface.py true_fused_attn_with_kvcache): `prefill_call` and `decode_call` are callables that each issue
    exactly ONE attention call of this module (flash_attn_with_kvcache / flash_attn_varlen_with_kvcache, results via out=); the
    two calls are recorded instead of launched and go to the GPU as one launch on the current stream."""
    global _capture
    if _capture is not None:
        raise RuntimeError("hybrid_attn does not nest")
    _capture = []
    try:
        prefill_call()
        n_pre = len(_capture)
        decode_call()
        cap = _capture
    finally:
        _capture = None
    if n_pre != 1 or len(cap) != 2:
        raise RuntimeError("hybrid_attn: each part must issue exactly one attention call (got %d + %d)" % (n_pre, len(cap) - n_pre))
    (pp, keep_p), (pd, keep_d) = cap
    if _role_mode:
        pp.variant = (pp.variant & ~(3 << 12)) | ((_role_mode & 3) << 12)
    # the fused launch is lab-only since round 4 (csrc/hybrid_kernels.hip: measured slower than serial three ways); _product: the product
    # library's entry point of the same name, which issues the two launches back to back
    lib = K.klib() if _product else K.klib_lab()
    need = lib.vattn_hybrid_workspace_bytes(C.byref(pp), C.byref(pd))
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _hybrid_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.zeros((max(need, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)      # zero ONCE: see the header
        _hybrid_ws[key] = ws
    rc = lib.vattn_hybrid_attn(C.byref(pp), C.byref(pd), ws.data_ptr(), K.current_stream_ptr(device))
    if rc != 0:
        raise RuntimeError(K.last_error(lib))
    del keep_p, keep_d


def _capture_active() -> bool:
    return _capture is not None


def relaunch(p, q_ptr: int, k_new_ptr: int, v_new_ptr: int, out_ptr: int, k_cache_ptr: int, v_cache_ptr: int, dev) -> None:
    """Re-issue a call whose parameter block `p` was built by flash_attn_with_kvcache(..., _params_out=[]) with other tensors of the
    SAME shapes, strides and dtype (the attention wrapper: layers 1..L-1 of an iteration).  Only the six data pointers change; the
    workspace is looked up again (another call on this stream may have replaced it with a larger one)."""
    p.q, p.out, p.k_cache, p.v_cache = q_ptr, out_ptr, k_cache_ptr, v_cache_ptr
    if p.k_new:
        p.k_new, p.v_new = k_new_ptr, v_new_ptr
    fast = getattr(p, "_fast", None)
    if fast is None or _capture is not None:
        _launch(p, dev)
        return
    # (the block's library and workspace need were settled by its first launch: a batch-1 decode is tens of microseconds per layer, and
    # every host microsecond on this path shows up as an idle GPU)
    lib, need = fast
    st = K.current_stream_ptr(dev)
    if need:
        p.workspace = _workspace(need, dev, st).data_ptr()
    rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
    if rc != 0:
        raise RuntimeError(K.last_error(lib))


def _launch(p, dev, keep=()):
    """Attach the split-KV workspace the call needs (one buffer per device and stream) and launch on the current stream."""
    if _capture is not None:
        _capture.append((p, keep))
        return
    lib = K.klib_for(p.variant)          # the product library; the lab build only for measurement variants (tests, kbench)
    need = lib.vattn_attn_workspace_bytes(C.byref(p))
    st = K.current_stream_ptr(dev)
    if need:
        ws = _workspace(need, dev, st)   # kept alive by the per-(device, stream) cache until a larger one replaces it
        p.workspace = ws.data_ptr()
    rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
    if rc != 0:
        raise RuntimeError(K.last_error(lib))
    p._fast = (lib, need)                # relaunch(): same shapes, same plan


def _check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vattention_amd.flash_attn: tensors must live on the GPU (there is no CPU path)")


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens: Optional[Union[int, torch.Tensor]] = None,
                            cache_batch_idx: Optional[torch.Tensor] = None, cache_leftpad=None, block_table=None,
                            softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                            rotary_interleaved=True, alibi_slopes=None, num_splits=0, return_softmax_lse=False,
                            out=None, _variant=0, _max_seqlen_k: int = 0, _rotary_cos_sin=None, _params_out=None,
                            _cache_seqlens_host=None, _plan_tiles: int = 0, _pf_plan=None):
    rot = _rotary_table(rotary_cos, rotary_sin, _rotary_cos_sin, rotary_interleaved, q)
    if block_table is not None:
        raise NotImplementedError("paged KV (block_table) is what vAttention replaces; not supported")
    if alibi_slopes is not None or cache_leftpad is not None or tuple(window_size) != (-1, -1) or softcap != 0.0:
        raise NotImplementedError("alibi / leftpad / sliding window / softcap are not used by the vAttention path")
    _check_cuda(q, k_cache, v_cache, k, v)
    assert k_cache.stride(-1) == 1, "k_cache must have contiguous last dimension"
    assert v_cache.stride(-1) == 1, "v_cache must have contiguous last dimension"
    mc = lambda x: x.contiguous() if x is not None and x.stride(-1) != 1 else x
    q, k, v = mc(q), mc(k), mc(v)
    B, Sq, Hq, D = q.shape
    Bc, Sk, Hkv, Dk = k_cache.shape
    if k_cache.dtype != q.dtype:
        raise RuntimeError("query and key must have the same dtype")
    if v_cache.dtype != q.dtype:
        raise RuntimeError("query and value must have the same dtype")
    if softmax_scale is None:
        softmax_scale = D ** (-0.5)
    dev = q.device
    # host-side bound on the sequences' lengths (sizes the prefill KV split): known when cache_seqlens is an int or the
    # caller (the attention wrapper, which has the lengths on the host) passes _max_seqlen_k; else the cache's row count
    hint = int(_max_seqlen_k)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        hint = hint or cache_seqlens + (k.shape[1] if k is not None else 0)
        cache_seqlens = torch.full((B,), cache_seqlens, dtype=torch.int32, device=dev)
    if cache_seqlens is not None:
        if cache_seqlens.dtype != torch.int32:
            raise RuntimeError("seqlens_k must have dtype int32")
        cache_seqlens = cache_seqlens.contiguous()
        assert cache_seqlens.shape == (B,)
    if cache_batch_idx is not None:
        if cache_batch_idx.dtype != torch.int32:
            raise RuntimeError("cache_batch_idx must have dtype int32")
        cache_batch_idx = cache_batch_idx.contiguous()
    elif Bc < B:
        raise RuntimeError("batch size of the cache is smaller than the batch size of q")
    Sn = 0
    if k is not None:
        if v is None:
            raise RuntimeError("If key is supplied, value must also be passed in")
        if cache_seqlens is None:
            raise RuntimeError("If key is supplied, seqlens_k must also be passed in")
        Sn = k.shape[1]
        if Sn > Sk:
            raise RuntimeError(APPEND_ERR)
        assert k.shape == (B, Sn, Hkv, D) and v.shape == (B, Sn, Hkv, D)
    if out is None:        # `out` mirrors the optional out_ of the reference's C++ entry point (flash_api.cpp:1303)
        out = torch.empty_like(q)
    else:
        if out.dtype != q.dtype:
            raise RuntimeError("Output must have the same dtype as inputs")
        if out.shape != q.shape or out.stride(-1) != 1:
            raise RuntimeError("Output tensor must have the shape of q and a contiguous last dimension")
    lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=dev) if return_softmax_lse else None

    p = K.AttnParams()
    p.q, p.out = q.data_ptr(), out.data_ptr()
    p.q_batch_stride, p.q_row_stride, p.q_head_stride = q.stride(0), q.stride(1), q.stride(2)
    p.o_batch_stride, p.o_row_stride, p.o_head_stride = out.stride(0), out.stride(1), out.stride(2)
    _set_cache(p, k_cache, v_cache)
    if k is not None:
        p.k_new, p.v_new = k.data_ptr(), v.data_ptr()
        p.knew_batch_stride, p.knew_row_stride, p.knew_head_stride = k.stride(0), k.stride(1), k.stride(2)
        p.vnew_batch_stride, p.vnew_row_stride, p.vnew_head_stride = v.stride(0), v.stride(1), v.stride(2)
    p.cache_seqlens = cache_seqlens.data_ptr() if cache_seqlens is not None else None
    p.cache_batch_idx = cache_batch_idx.data_ptr() if cache_batch_idx is not None else None
    p.softmax_lse = lse.data_ptr() if lse is not None else None
    p.b, p.seqlen_q, p.seqlen_k, p.seqlen_knew, p.h, p.h_k, p.d = B, Sq, Sk, Sn, Hq, Hkv, D
    p.is_causal = 1 if causal else 0
    p.dtype = K.dtype_code(q.dtype)
    p.num_splits = int(num_splits)
    p.softmax_scale = float(softmax_scale)
    p.variant = int(_variant)
    p.split_reserved = _STREAM_SWITCH
    # Prefill form WITHOUT any host-side length (the reference's own call: vattention_flashattention_wrapper.py:159-166 passes the slot's
    # whole row-block and `cache_seqlens` as a device tensor): the `vattention` drop-in was told every slot's length of this iteration in
    # step_async(seq_lens) — resolve the row-block's address to its slot and take the length from there.  No device-to-host copy, no
    # extra argument; a tensor that is not one of the page manager's leaves the view's row count as the bound (FlashAttention's own rule).
    klens = _cache_seqlens_host
    if USE_PAGE_MANAGER_LENGTHS and Sq > 1 and hint == 0 and klens is None and cache_seqlens is not None and cache_batch_idx is None and k is None:
        klens = _lengths_from_page_manager(k_cache, B)
        if klens is not None:
            hint = max(klens)
            counters["lengths_from_page_manager"] += 1
    p.max_seqlen_k_hint = min(hint, Sk + Sn) if hint > 0 else 0
    if rot is not None:
        p.rotary_cos_sin, p.rotary_row_stride, p.rotary_dim = rot.data_ptr(), rot.stride(0), rot.shape[1]
    plan = None
    if Sq > 1 and D == 128 and num_splits == 0 and k is None and not _capture_active():
        # prefill form: a work list for underfilled / unbalanced grids.  `_pf_plan`: a plan object built earlier for the same lengths
        # (this package's wrapper: one per iteration), else built here — and kept, keyed on the shapes and lengths: the L layers of an
        # iteration issue the same call — from the host-side lengths when there are any (_cache_seqlens_host, or the page manager's).
        # The list is a performance hint only: its ranges are clamped to the device-side lengths (include/vattn_kernels.h).
        if isinstance(_pf_plan, _PrefillPlan):
            plan = _pf_plan
        elif klens is not None and (_pf_plan == "host" or _pf_plan is None) and not torch.cuda.is_current_stream_capturing():
            plan = _cached_prefill_plan(p, klens, dev)
        if plan is not None:
            plan.attach(p)
            counters["work_list_attached"] += plan.t is not None
    if Sq > 1:
        counters["prefill_calls"] += 1
    if _cache_seqlens_host is not None and Sq == 1 and B > 1 and num_splits == 0 and not torch.cuda.is_current_stream_capturing():
        # (the plan's tables travel by a host-to-device copy: not while the stream is being captured into a graph — the uniform split then)
        if _plan_tiles:                                        # (tests / A-B: pieces of exactly this many 32-key tiles)
            p.num_splits = -int(_plan_tiles)
        plan = _decode_plan(p, _cache_seqlens_host, dev)      # ragged batch: work items of near-equal length (None: uniform split)
        p.num_splits = 0
    _launch(p, dev, keep=(q, k, v, k_cache, v_cache, cache_seqlens, cache_batch_idx, out, lse, rot, plan))
    if _params_out is not None and not return_softmax_lse:
        # the caller may re-issue this call through relaunch(): the block keeps the index / length / rotary tensors it points to alive
        p._keep = (cache_seqlens, cache_batch_idx, rot, plan)
        _params_out.append(p)
    return (out, lse) if return_softmax_lse else out


def _lengths_from_page_manager(k_cache, B: int):
    """Visible tokens of the B row-blocks of `k_cache` according to the page manager's last step (vattention.resolve_view), or None."""
    from . import vattention as _va
    if _va._pm is None:
        return None
    ptr, step = k_cache.data_ptr(), k_cache.stride(0) * k_cache.element_size()
    out = []
    for b in range(B):
        r = _va.resolve_view(ptr + b * step)
        if r is None or r[1] > k_cache.shape[1]:
            return None
        out.append(r[1])
    return out


_plan_cache = {}      # (shapes, lengths, device, stream) -> _PrefillPlan; a few dozen entries, dropped wholesale when full
# Prefill calls WITHOUT a host-side length take it from the page manager's last step (vattention.resolve_view).  The
# lengths only size launch plans — the kernels clamp to cache_seqlens on the device — but a caller that steps the manager and attends
# with other lengths in between gets plans sized for the wrong lengths; False switches the lookup off (the view's row count then bounds
# the plan, FlashAttention's own rule).
USE_PAGE_MANAGER_LENGTHS = True
counters = {"prefill_calls": 0, "lengths_from_page_manager": 0, "plan_built": 0, "plan_cache_hit": 0, "work_list_attached": 0}      # introspection (tools/, tests)


def _cached_prefill_plan(p, klens, dev):
    # keyed on the STREAM too: the tables travel by a copy queued on the stream current at build time, and nothing orders a launch on
    # another stream behind that copy
    # ... and on what decides the FORM of the list: the persistent-queue switches, fused RoPE and the output stride alignment
    # (both force one workgroup per piece: prefill_plan) — a list built under one policy is not reused under another
    key = (p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal, tuple(klens), dev.index, torch.cuda.current_stream(dev).cuda_stream,
           PERSISTENT, PERSISTENT_DRAWN, bool(p.rotary_cos_sin), bool((p.o_row_stride | p.o_head_stride | p.o_batch_stride) & 7))
    pl = _plan_cache.get(key)
    if pl is None:
        if len(_plan_cache) >= 64:
            _plan_cache.clear()
        pl = _plan_cache[key] = prefill_plan(p, None, klens, dev)
        counters["plan_built"] += 1
    else:
        counters["plan_cache_hit"] += 1
    return pl


class _Staging:
    """Pinned host buffers for the small plan tables: a pageable host-to-device copy is staged synchronously by the runtime (the host
    thread waits behind whatever the stream has queued); from pinned memory the copy is queued like a launch.  A ring of 16 buffers, each
    guarded by the event of its last copy."""
    ring, events, pos = [], [], 0

    @classmethod
    def upload(cls, nbytes: int, fill, dev) -> torch.Tensor:
        if not cls.ring:
            cls.ring = [torch.empty(1 << 16, dtype=torch.uint8).pin_memory() for _ in range(16)]
            cls.events = [None] * 16
        i = cls.pos = (cls.pos + 1) % 16
        if cls.ring[i].numel() < nbytes:
            cls.ring[i] = torch.empty(max(nbytes, 2 * cls.ring[i].numel()), dtype=torch.uint8).pin_memory()
            cls.events[i] = None
        if cls.events[i] is not None:
            cls.events[i].synchronize()
        buf = cls.ring[i]
        fill(buf.data_ptr())
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        t.copy_(buf[:nbytes], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        cls.events[i] = ev
        return t


class _PrefillPlan:
    """Device tables of a prefill work list (vattn_prefill_plan) — or the fact that the default launch is as good (tables None).  Built
    from host-side lengths only, so the attention wrapper builds it for layer 0 of an iteration and hands the same object to the
    other layers' calls (`_pf_plan`)."""
    __slots__ = ("t", "n_items", "n_blocks", "part_rows", "n_wg", "drawn")

    def __init__(self, t=None, n_items=0, n_blocks=0, part_rows=0, n_wg=0, drawn=False):
        self.t, self.n_items, self.n_blocks, self.part_rows, self.n_wg, self.drawn = t, n_items, n_blocks, part_rows, n_wg, drawn

    def attach(self, p):
        if self.t is not None:
            base = self.t.data_ptr()
            p.pf_items, p.num_pf_items = base, self.n_items
            p.pf_blocks, p.num_pf_blocks = (base + 32 * self.n_items if self.n_blocks else None), self.n_blocks
            p.pf_part_rows = self.part_rows
            # persistent form (vattn_prefill_plan_wg): the pieces are grouped by workgroup, the offsets follow the two tables
            p.pf_num_wg = self.n_wg
            p.pf_wg_first = base + 32 * (self.n_items + self.n_blocks) if self.n_wg and not self.drawn else None


# Work lists walked by PERSISTENT workgroups (csrc/prefill64p_kernels.hip) — where that measured faster on every box it was tried on
# (profiles/r05_p64p_kbench_ab.txt, r05_p64p_legs_ab_*.txt): ONE entry, a chunk on a prefix of at least 4 x its length (Sarathi chunks:
# equal pieces, several per workgroup: +1 ... +4 %).  Whole prompts and ragged batches keep one workgroup per piece: there the host-assigned
# queues measure EQUAL to it, alone and inside the replay (TP8-rank leg 0.411 always / 0.413 this policy / 0.412 never,
# profiles/r05_timer_stride.txt), the drawn queues 8-13 % slower alone, and the assignment costs host time in front of layer 0's launch.
# "always" / "never": every list / none (tests, tools/kbench.py, bench.py --persistent-prefill / --per-piece-prefill).
PERSISTENT = "chunks"
PERSISTENT_DRAWN = False   # True: the workgroups DRAW their pieces from a device counter; False: host-assigned queues (measured better: see above)
PERSISTENT_MAX_BLOCKS = 2048      # (entry, head, query block) triples up to which a launch gets a list at all in the persistent form


def prefill_plan(p, q_lens_host, k_lens_host, dev, force_tiles: int = 0, persistent=None, max_wg: int = 0, drawn=None) -> _PrefillPlan:
    """q_lens_host: chunk length per entry (None: p.seqlen_q for all); k_lens_host: visible keys per entry; force_tiles: pieces of at
    most this many 64-key tiles whatever the planner's own rules say (tests, A/B); persistent: None = the module default; max_wg: at
    most this many persistent workgroups (0: one per CU)."""
    B = p.b
    if persistent is None:
        q0 = int(q_lens_host[0]) if q_lens_host is not None else p.seqlen_q
        persist = PERSISTENT == "always" or (PERSISTENT == "chunks" and B == 1 and int(k_lens_host[0]) >= 5 * q0)
    else:
        persist = bool(persistent)
    if p.rotary_cos_sin or (p.o_row_stride | p.o_head_stride | p.o_batch_stride) & 7:
        persist = False          # (the persistent kernel has neither the fused-RoPE form nor the 8-byte store path)
    q_of = q_lens_host if q_lens_host is not None else [p.seqlen_q] * B
    n_blk = sum((int(q) + 255) // 256 for q in q_of) * p.h          # (entry, head, 256-row query block) triples
    cap_i, cap_b = 17 * n_blk + 16, n_blk + 16
    ragged = q_lens_host is not None and len({(int(q) + 255) // 256 for q in q_lens_host}) > 1
    if persist and n_blk > PERSISTENT_MAX_BLOCKS and not force_tiles and not ragged:
        return _PrefillPlan()
    if not persist and n_blk > 4 * 256 + 64 and not force_tiles and not ragged:      # (the planner keeps the default launch for balanced grids of several rounds)
        return _PrefillPlan()
    items, blocks = (K.PrefillItem * cap_i)(), (K.PrefillItem * cap_b)()
    counts = (C.c_int32 * 3)()
    ql = (C.c_int32 * B)(*[int(x) for x in q_lens_host]) if q_lens_host is not None else None
    kl = (C.c_int32 * B)(*[int(x) for x in k_lens_host])
    keep_ns = p.num_splits
    if force_tiles:
        p.num_splits = -int(force_tiles)
    if persist:
        counts = (C.c_int32 * 4)()
        dyn = PERSISTENT_DRAWN if drawn is None else bool(drawn)
        wg_first = None if dyn else (C.c_int32 * 257)()
        n = K.klib().vattn_prefill_plan_wg(C.byref(p), ql, kl, items, cap_i, blocks, cap_b, wg_first, int(max_wg), counts)
    else:
        n = K.klib().vattn_prefill_plan(C.byref(p), ql, kl, items, cap_i, blocks, cap_b, counts)
    p.num_splits = keep_ns
    if n < 0:
        raise RuntimeError("vattn_prefill_plan: bad arguments")
    if n == 0:
        return _PrefillPlan()
    nb = int(counts[1])
    nwg = int(counts[3]) if persist else 0

    def fill(dst):
        C.memmove(dst, items, 32 * n)
        if nb:
            C.memmove(dst + 32 * n, blocks, 32 * nb)
        if nwg and wg_first is not None:
            C.memmove(dst + 32 * (n + nb), wg_first, 4 * (nwg + 1))
    return _PrefillPlan(_Staging.upload(32 * (n + nb) + (4 * (nwg + 1) if nwg and wg_first is not None else 0), fill, dev), n, nb, int(counts[2]), nwg,
                        drawn=bool(nwg and wg_first is None))


def _decode_plan(p, lens_host, dev):
    """Length-balanced split of a ragged decode batch (include/vattn_kernels.h, vattn_decode_plan): the host-side lengths -> two small
    device tables, attached to the parameter block.  One H2D copy; the attention wrapper builds it for layer 0 of an iteration and
    the other layers re-issue the same block."""
    B = p.b
    if len(lens_host) != B:
        raise RuntimeError("_cache_seqlens_host must have one entry per batch element")
    cap = 4 * B + 1024
    lens = (C.c_int32 * B)(*[int(x) for x in lens_host])
    items = (K.DecodeItem * cap)()
    seq = (C.c_int32 * (2 * B))()
    n = K.klib_for(p.variant).vattn_decode_plan(C.byref(p), lens, items, cap, seq)      # (the library that will run the call: its slot count)
    if n < 0:
        raise RuntimeError("vattn_decode_plan: bad arguments")
    if n == 0:
        return None
    def fill(dst):
        C.memmove(dst, items, 16 * n)
        C.memmove(dst + 16 * n, seq, 8 * B)
    t = _Staging.upload(16 * n + 8 * B, fill, dev)
    p.split_items, p.split_seq, p.num_split_items = t.data_ptr(), t.data_ptr() + 16 * n, n
    return t


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """flash_attn_interface.py flash_attn_func (forward only, no dropout): q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D]."""
    if dropout_p != 0.0 or return_attn_probs:
        raise NotImplementedError("dropout / attention probabilities are not supported (inference path)")
    return flash_attn_with_kvcache(q, k, v, softmax_scale=softmax_scale, causal=causal, window_size=window_size,
                                   softcap=softcap, alibi_slopes=alibi_slopes)


def flash_attn_varlen_with_kvcache(q, k_cache, v_cache, q_start: torch.Tensor, q_lens: torch.Tensor, max_q_len: int,
                                   cache_seqlens: torch.Tensor, cache_batch_idx: Optional[torch.Tensor] = None,
                                   softmax_scale=None, causal=True, out=None, num_splits=0, _variant=0, _max_seqlen_k: int = 0,
                                   _rotary_cos_sin=None, _pf_plan=None):
    """MI355X extension (SURVEY §8f "batched multi-prefill"): ONE launch for the prefill chunks of several sequences with
    different lengths.  q / out are the flattened tokens [T, Hq, D]; entry i attends with rows [q_start[i], q_start[i] +
    q_lens[i]) over cache slot cache_batch_idx[i] (identity if None), keys [0, cache_seqlens[i]) — the chunk's own K/V must
    already be in the cache (cache_flat).  Bottom-right-aligned causal mask per entry, exactly as flash_attn_with_kvcache
    does for one sequence (the reference's wrapper issues one call per prompt, vattention_flashattention_wrapper.py:129-174)."""
    _check_cuda(q, k_cache, v_cache, q_start, q_lens, cache_seqlens, cache_batch_idx)
    if q.dim() != 3:
        raise RuntimeError("q must be [total_tokens, num_heads, head_size]")
    T, Hq, D = q.shape
    Bc, Sk, Hkv, Dk = k_cache.shape
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise RuntimeError("query, key and value must have the same dtype")
    for t, name in ((q_start, "q_start"), (q_lens, "q_lens"), (cache_seqlens, "seqlens_k")):
        if t.dtype != torch.int32:
            raise RuntimeError(name + " must have dtype int32")
    B = q_lens.shape[0]
    assert q_start.shape == (B,) and cache_seqlens.shape == (B,)
    if cache_batch_idx is not None:
        if cache_batch_idx.dtype != torch.int32:
            raise RuntimeError("cache_batch_idx must have dtype int32")
        assert cache_batch_idx.shape == (B,)
    elif Bc < B:
        raise RuntimeError("batch size of the cache is smaller than the number of chunks")
    if max_q_len < 2:
        raise RuntimeError("max_q_len must be >= 2 (single-token queries take the decode form)")
    assert k_cache.stride(-1) == 1 and v_cache.stride(-1) == 1 and q.stride(-1) == 1
    if softmax_scale is None:
        softmax_scale = D ** (-0.5)
    if out is None:
        out = torch.empty_like(q)
    elif out.shape != q.shape or out.dtype != q.dtype or out.stride(-1) != 1:
        raise RuntimeError("Output tensor must have the shape and dtype of q and a contiguous last dimension")
    dev = q.device
    p = K.AttnParams()
    p.q, p.out = q.data_ptr(), out.data_ptr()
    p.q_batch_stride, p.q_row_stride, p.q_head_stride = 0, q.stride(0), q.stride(1)
    p.o_batch_stride, p.o_row_stride, p.o_head_stride = 0, out.stride(0), out.stride(1)
    _set_cache(p, k_cache, v_cache)
    p.cache_seqlens = cache_seqlens.contiguous().data_ptr()
    p.cache_batch_idx = cache_batch_idx.contiguous().data_ptr() if cache_batch_idx is not None else None
    p.q_start, p.q_lens = q_start.contiguous().data_ptr(), q_lens.contiguous().data_ptr()
    p.b, p.seqlen_q, p.seqlen_k, p.seqlen_knew, p.h, p.h_k, p.d = B, int(max_q_len), Sk, 0, Hq, Hkv, D
    p.is_causal = 1 if causal else 0
    p.dtype = K.dtype_code(q.dtype)
    p.num_splits = int(num_splits)
    p.softmax_scale = float(softmax_scale)
    p.variant = int(_variant)
    p.max_seqlen_k_hint = min(int(_max_seqlen_k), Sk) if _max_seqlen_k > 0 else 0
    if _rotary_cos_sin is not None:      # q rows of entry i are rotated at positions (cache_seqlens[i] - q_lens[i]) + row
        rot = _rotary_table(None, None, _rotary_cos_sin, False, q)
        p.rotary_cos_sin, p.rotary_row_stride, p.rotary_dim = rot.data_ptr(), rot.stride(0), rot.shape[1]
    if isinstance(_pf_plan, _PrefillPlan) and D == 128 and num_splits == 0 and not _capture_active():
        _pf_plan.attach(p)        # work list built from the host-side lengths of this iteration (prefill_plan)
    _launch(p, dev, keep=(q, k_cache, v_cache, q_start, q_lens, cache_seqlens, cache_batch_idx, out, _rotary_cos_sin, _pf_plan))
    return out
