"""Drop-in for the reference's `vattention` Python module.

Same 13 module-level functions on a process-global allocator, same argument order, return types
and exceptions (/root/reference/vattention/vattention.cu:614-637, apis.h:1-63; callers:
/root/reference/sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py:44-191).
Everything is forwarded to the C ABI of libvattn_amd.so (include/vattn.h): HIP VMM backend,
mapper thread, no fallback.  `release_kvcache_physical` is called by the reference engine
(vATTN_cache_engine.py:164-165) but never exported by the reference module; it is a no-op here.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib as L
from .page_manager import PageManager

_pm: Optional[PageManager] = None
_tensors: List[torch.Tensor] = []
_bases: List[int] = []         # base address of every virtual tensor (resolve_view)
_last_lens: List[int] = []     # the seq_lens of the last step()/step_async(): slot -> visible tokens of this iteration
_verbose = False
_deferred = True
_default_flags = 0
_layered_pending = False     # a layer-ordered mapping batch of the current step may still be running (see wait_layer)


def _vtensor():
    from . import _vtensor as ext      # torch binding built by vattention_amd/build.py; fails loudly if absent
    return ext


def _require() -> PageManager:
    if _pm is None:
        raise RuntimeError("vattention: init_kvcache() has not been called")
    return _pm


def set_backend_flags(flags: int) -> None:
    """MI355X extension: VATTN_FLAG_* bits applied by the next init_kvcache (see include/vattn.h)."""
    global _default_flags
    _default_flags = int(flags)


def init_kvcache(num_layers: int, num_kv_heads: int, head_size: int, max_batch_size: int,
                 max_context_length: int, device: int, dtype: torch.dtype, page_size: int,
                 megacache: bool) -> List[torch.Tensor]:
    """apis.h:3-13.  Returns 2*L virtual tensors [B, max_ctx, kvh, D] (K_0..K_{L-1}, V_0..V_{L-1}) —
    or two [B, max_ctx, L, kvh, D] tensors with megacache — on cuda:<device>, with NO physical
    memory behind them yet.  A HIP context must exist (the reference says "initialize PyTorch first")."""
    global _pm, _tensors, _bases, _last_lens
    if _pm is not None:
        cleanup()
    itemsize = torch.empty((), dtype=dtype).element_size()
    torch.cuda.set_device(device)
    torch.cuda.current_stream(device)          # make sure the primary context is live on this thread
    pm = PageManager(num_layers, num_kv_heads, head_size, max_batch_size, max_context_length, itemsize,
                     device, page_size, bool(megacache), flags=_default_flags, backend=None)
    pm.set_verbose(_verbose)
    pm.set_deferred_reclamation(_deferred)
    ext = _vtensor()
    shape, stride = pm.shape(), pm.stride()
    tensors = [ext.tensor_from_va(pm.tensor_base(i), shape, stride, dtype, device) for i in range(pm.num_tensors)]
    _pm, _tensors = pm, tensors
    _bases = [int(pm.tensor_base(i)) for i in range(pm.num_tensors)]
    _last_lens = [0] * max_batch_size
    return list(tensors)


_wait_pool = True
POOL_READY_TIMEOUT_MS = 600_000     # a 259 GB pool of 2 MiB pages is 98-125 s of hipMemCreate; never wait forever on a wedged mapper
pool_ready_seconds = 0.0       # what the last reserve_physical_pages spent waiting for the pool's handles (introspection: bench.py)


def set_wait_pool_ready(on: bool) -> None:
    """MI355X extension.  True (default): reserve_physical_pages returns once the mapper thread has created the handles it creates ahead
    of demand — the whole pool up to 40 000 pages (1-1.5 s for 31 k handles of 8 MiB), a 4 096-handle window above that — which is
    what the reference's reserve does for every page (cudaInternal.h:45-59).  False: return at once and let creation run under the
    first iterations (their launches then wait for the driver behind hipMemCreate: profiles/r05_cold_pool.md)."""
    global _wait_pool
    _wait_pool = bool(on)


def reserve_physical_pages(free_memory: int) -> int:
    """apis.h:23-25: number of physical pages in the pool (multiple of 2*L)."""
    global pool_ready_seconds
    n = _require().reserve_physical_pages(free_memory)
    if _wait_pool:
        import time
        t0 = time.perf_counter()
        left = _require().wait_pool_ready(POOL_READY_TIMEOUT_MS)
        pool_ready_seconds = time.perf_counter() - t0
        if left < 0:      # the reference aborts inside reserve when the pool cannot be committed (cudaInternal.h:45-59, CHECK_CUDA)
            raise RuntimeError("reserve_physical_pages: creating the pool's physical handles failed (error %d): the pool does not fit the device" % left)
        if left > 0:
            raise RuntimeError("reserve_physical_pages: the mapper thread created no handle for %.0f s (%d left): wedged driver?" % (POOL_READY_TIMEOUT_MS / 1e3, left))
    return n


def step(seq_lens: List[int], eager_reclaim: bool) -> None:
    """apis.h:27-29 (the `_sync` backends)."""
    global _last_lens
    _require().step(seq_lens, eager_reclaim)
    _last_lens = list(seq_lens)


def step_async(seq_lens: List[int]) -> None:
    """apis.h:31-35: maps what this iteration needs before returning, plans and hands the look-ahead
    mapping to the mapper thread; the GIL is released for the duration of the native call."""
    global _layered_pending, _last_lens
    pm = _require()
    pm.step_async(seq_lens)
    _last_lens = list(seq_lens)
    if pm.cfg.flags & L.FLAG_LAYERED_ASYNC:
        _layered_pending = pm.layers_ready() < pm.cfg.num_layers


def alloc_new_batch_idx(seqlen: int) -> int:
    return _require().alloc_new_batch_idx(seqlen)


def free_batch_idx(reqId: int) -> None:
    """apis.h:57-59.  Also marks, on torch's current stream, the point after the kernels launched so far: a later reclaim of the
    slot's pages waits for that point instead of synchronising the device (include/vattn.h, vattn_free_batch_idx_on_stream)."""
    pm = _require()
    pm.free_batch_idx(reqId, stream=torch.cuda.current_stream(pm.cfg.device).cuda_stream)


def num_free_kvblocks() -> int:
    return _require().num_free_kvblocks()


def cleanup() -> None:
    """apis.h:41-43: joins the mapper, unmaps everything, frees VA and physical handles.  Tensors
    returned by init_kvcache must not be used afterwards."""
    global _pm, _tensors, _bases, _last_lens
    if _pm is None:
        return
    torch.cuda.synchronize()
    _pm.cleanup()
    _pm.close()
    _pm, _tensors, _bases, _last_lens = None, [], [], []


def set_verbose(val: bool) -> None:
    global _verbose
    _verbose = bool(val)
    if _pm is not None:
        _pm.set_verbose(_verbose)


def set_deferred_reclamation(val: bool) -> None:
    global _deferred
    _deferred = bool(val)
    if _pm is not None:
        _pm.set_deferred_reclamation(_deferred)


def show_kvcache_config() -> None:
    _require().show_kvcache_config()


def show_allocator_state() -> None:
    _require().show_allocator_state()


def map_common_pages(num_tokens: int) -> None:
    _require().map_common_pages(num_tokens)


def release_kvcache_physical() -> None:
    return None


# ---- MI355X extensions (not in the reference surface) ----
def enable_layered_async(on: bool = True) -> None:
    """The next init_kvcache maps new prompts LAYER-ORDERED (VATTN_FLAG_LAYERED_ASYNC): step_async returns after the first
    layers' pages are mapped, the mapper thread maps the rest while those layers run.  The caller MUST then call
    wait_layer(layer_id) before launching each layer (the fa_vattn wrapper of this package does; the reference's does not, so
    this stays opt-in)."""
    global _default_flags
    _default_flags = (_default_flags | L.FLAG_LAYERED_ASYNC) if on else (_default_flags & ~L.FLAG_LAYERED_ASYNC)


def premap(seqlen: int) -> int:
    """Admission look-ahead (include/vattn.h, vattn_premap): reserve the slot the NEXT request will get and map its pages on the
    mapper thread while the current iteration runs.  Returns the slot, -1 if none is free."""
    return _require().premap(seqlen)


def cancel_premap(slot: int) -> None:
    _require().cancel_premap(slot)


def wait_layer(layer_id: int) -> None:
    """Block until the pages the current step needs are mapped for `layer_id` (no-op unless a layered batch is pending)."""
    global _layered_pending
    if not _layered_pending or _pm is None:
        return
    _pm.wait_layer(layer_id)
    if layer_id + 1 >= _pm.cfg.num_layers:
        _layered_pending = False



def wait_all_layers() -> None:
    """Block until EVERY layer's pages of the current step are mapped.  What a caller that does not know its layer index must use
    (the reference's yi / mistral / qwen / falcon / internlm models call wrapper.forward() with layer_id=None; only llama.py:179-186
    passes it): the first such call of an iteration pays the whole wait, the following ones return at once."""
    global _layered_pending
    if not _layered_pending or _pm is None:
        return
    _pm.wait_layer(_pm.cfg.num_layers - 1)
    _layered_pending = False


def wait() -> None:
    """Join outstanding background mapping (the next step()/step_async() does this implicitly)."""
    _require().wait()


def stats() -> dict:
    return _require().stats()


def counts() -> dict:
    return _require().counts()


def resolve_view(ptr: int):
    """(slot, visible tokens) of the KV-cache row-block that starts at device address `ptr`, or None when `ptr` is not the first row of
    a slot of one of this manager's virtual tensors.  The attention drop-in uses it for the reference's PREFILL call
    (vattention_flashattention_wrapper.py:146-166): the wrapper hands over `kv_cache[l][slot]` — every row of the slot, max_ctx of them —
    and the length only as a DEVICE tensor, but the engine told this module every slot's length of the iteration in
    step_async(seq_lens) (vATTN_cache_engine.py:98-121): the launch plan needs no device-to-host copy and no extra argument."""
    if _pm is None or not _bases:
        return None
    lay = _pm.layout
    per_req = int(lay.virt_bytes_per_req)
    total = int(lay.virt_bytes_total)
    for base in _bases:
        off = ptr - base
        if 0 <= off < total:
            slot = off // per_req
            within = off - slot * per_req
            # the first row of the slot: offset 0, or layer l's rows of a megacache tensor ([B, ctx, L, kvh, D]: l * kvh * D * itemsize)
            row_bytes = _pm.cfg.num_kv_heads * _pm.cfg.head_size * _pm.cfg.itemsize
            if within >= (_pm.cfg.num_layers * row_bytes if _pm.cfg.megacache else 1):
                return None
            n = int(_last_lens[slot]) if slot < len(_last_lens) else 0
            return (int(slot), n) if n > 0 else None
    return None


def layout() -> dict:
    pm = _require()
    lay = pm.layout
    return {"shape": pm.shape(), "stride": pm.stride(), "virt_bytes_per_req": int(lay.virt_bytes_per_req),
            "virt_bytes_total": int(lay.virt_bytes_total), "tokens_per_page": int(lay.tokens_per_page),
            "max_pages_per_req": int(lay.max_pages_per_req), "page_size": int(lay.page_size)}


def state() -> dict:
    return _require().state()


def granularity(device: int = 0):
    import ctypes as C
    a, b = C.c_uint64(), C.c_uint64()
    rc = L.lib().vattn_hip_granularity(device, C.byref(a), C.byref(b))
    if rc != 0:
        raise RuntimeError("HIP VMM granularity query failed")
    return int(a.value), int(b.value)
