"""ctypes loader for libvattn_amd.so (the C-ABI product library: include/vattn.h, include/vattn_kernels.h).

There is no fallback: if the shared library is missing the import fails loudly and tells the user
to build it (python -c "import __graft_entry__ as g; g.build()").
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvattn_amd.so")


class VattnConfig(C.Structure):
    _fields_ = [("num_layers", C.c_uint32), ("num_kv_heads", C.c_uint32), ("head_size", C.c_uint32),
                ("max_batch_size", C.c_uint32), ("max_context_length", C.c_uint64), ("itemsize", C.c_uint32),
                ("device", C.c_int32), ("page_size", C.c_uint64), ("megacache", C.c_uint32), ("flags", C.c_uint32)]


class VattnLayout(C.Structure):
    _fields_ = [("ndim", C.c_uint32), ("shape", C.c_uint64 * 5), ("stride", C.c_uint64 * 5),
                ("virt_bytes_per_req", C.c_uint64), ("virt_bytes_total", C.c_uint64),
                ("tokens_per_page", C.c_uint64), ("max_pages_per_req", C.c_uint64), ("page_size", C.c_uint64)]


class VattnStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("handles_created", "handles_released", "map_calls", "access_calls",
                                           "unmap_calls", "sync_batches", "async_batches", "sync_ns", "async_ns",
                                           "join_wait_ns", "create_ns", "pages_mapped_now", "tlb_flushes", "tlb_flush_ns", "quiesce_calls", "quiesce_ns",
                                           "fence_waits", "fence_wait_ns", "layered_batches", "layer_wait_ns", "rollbacks",
                                           "sync_create_ns", "sync_creates", "sync_fence_ns", "sync_tlb_ns", "sync_maps", "sync_unmaps")]


VATTN_OK, VATTN_ERR_INVALID, VATTN_ERR_OOM, VATTN_ERR_DRIVER, VATTN_ERR_POOL_EMPTY = 0, -1, -2, -3, -4
FLAG_EAGER_CREATE, FLAG_NO_ACCESS_MERGE, FLAG_NO_MAPPER_THREAD, FLAG_LAYERED_ASYNC, FLAG_NO_VMM_SELFCHECK = 1, 2, 4, 8, 16

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "vattention_amd: %s is missing — the HIP extension has not been built. "
            "Run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root. "
            "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, i64, u32, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_int
    sig = {
        "vattn_create": (i32, [C.POINTER(VattnConfig), vp, C.POINTER(vp)]),
        "vattn_num_tensors": (i32, [vp]),
        "vattn_tensor_base": (u64, [vp, i32]),
        "vattn_get_layout": (i32, [vp, C.POINTER(VattnLayout)]),
        "vattn_reserve_physical_pages": (i64, [vp, u64]),
        "vattn_step": (i32, [vp, C.POINTER(u64), u32, i32]),
        "vattn_step_async": (i32, [vp, C.POINTER(u64), u32]),
        "vattn_wait": (i32, [vp]),
        "vattn_alloc_new_batch_idx": (i32, [vp, u64]),
        "vattn_free_batch_idx": (i32, [vp, i32]),
        "vattn_free_batch_idx_on_stream": (i32, [vp, i32, vp]),
        "vattn_premap": (i32, [vp, u64]),
        "vattn_wait_pool_ready": (i64, [vp, i64]),
        "vattn_cancel_premap": (i32, [vp, i32]),
        "vattn_wait_layer": (i32, [vp, u32]),
        "vattn_layers_ready": (u32, [vp]),
        "vattn_set_sync_layers": (i32, [vp, u32]),
        "vattn_vmm_selfcheck": (i32, [i32, C.POINTER(u32)]),
        "vattn_num_free_kvblocks": (u64, [vp]),
        "vattn_set_deferred_reclamation": (i32, [vp, i32]),
        "vattn_set_verbose": (i32, [vp, i32]),
        "vattn_map_common_pages": (i32, [vp, u64]),
        "vattn_show_kvcache_config": (i32, [vp]),
        "vattn_show_allocator_state": (i32, [vp]),
        "vattn_cleanup": (i32, [vp]),
        "vattn_destroy": (None, [vp]),
        "vattn_state_dump": (i64, [vp, C.POINTER(u64), u64]),
        "vattn_pagemap_dump": (i64, [vp, C.POINTER(u64), u64]),
        "vattn_get_stats": (i32, [vp, C.POINTER(VattnStats)]),
        "vattn_get_counts": (i32, [vp, C.POINTER(u64)]),
        "vattn_last_error": (C.c_char_p, [vp]),
        "vattn_hip_granularity": (i32, [i32, C.POINTER(u64), C.POINTER(u64)]),
        "vattn_hip_versions": (i32, [C.POINTER(i32), C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
