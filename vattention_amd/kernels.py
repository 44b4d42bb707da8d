"""ctypes binding of the kernel C ABI (include/vattn_kernels.h)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32


ABI_VERSION = 5      # VATTN_KERNELS_ABI of include/vattn_kernels.h


class AttnParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("q", vp), ("out", vp),
        ("q_batch_stride", i64), ("q_row_stride", i64), ("q_head_stride", i64),
        ("o_batch_stride", i64), ("o_row_stride", i64), ("o_head_stride", i64),
        ("k_cache", vp), ("v_cache", vp),
        ("k_batch_stride", i64), ("k_row_stride", i64), ("k_head_stride", i64),
        ("v_batch_stride", i64), ("v_row_stride", i64), ("v_head_stride", i64),
        ("k_new", vp), ("v_new", vp),
        ("knew_batch_stride", i64), ("knew_row_stride", i64), ("knew_head_stride", i64),
        ("vnew_batch_stride", i64), ("vnew_row_stride", i64), ("vnew_head_stride", i64),
        ("cache_seqlens", vp), ("cache_batch_idx", vp), ("softmax_lse", vp), ("workspace", vp),
        ("q_start", vp), ("q_lens", vp),
        ("b", i32), ("seqlen_q", i32), ("seqlen_k", i32), ("seqlen_knew", i32), ("h", i32), ("h_k", i32), ("d", i32),
        ("is_causal", i32), ("dtype", i32), ("num_splits", i32), ("softmax_scale", C.c_float), ("variant", i32),
        ("max_seqlen_k_hint", i32),
        ("rotary_cos_sin", vp), ("rotary_row_stride", i64), ("rotary_dim", i32), ("rotary_reserved", i32),
        ("split_items", vp), ("split_seq", vp), ("num_split_items", i32), ("split_reserved", i32),
        ("pf_items", vp), ("pf_blocks", vp), ("num_pf_items", i32), ("num_pf_blocks", i32), ("pf_part_rows", i32), ("pf_num_wg", i32), ("pf_wg_first", vp),
    ]


def _attn_params_init(self, *a, **kw):
    C.Structure.__init__(self, *a, **kw)      # zero-initialised by ctypes
    self.struct_size, self.abi_version = C.sizeof(AttnParams), ABI_VERSION


AttnParams.__init__ = _attn_params_init


class PlanDesc(C.Structure):
    _fields_ = [("form", i32), ("path", i32), ("tiling", i32), ("nsplit", i32), ("workgroups", i32), ("merge_launch", i32), ("workspace_bytes", i64)]


class PrefillItem(C.Structure):
    _fields_ = [("b", i32), ("h", i32), ("qb", i32), ("tile_begin", i32), ("tile_end", i32), ("nshares", i32), ("part_row", i32), ("reserved", i32)]


class DecodeItem(C.Structure):
    _fields_ = [("b", i32), ("tile_begin", i32), ("tile_end", i32), ("index_in_seq", i32)]


_bound = False
_lab = None
# variant bits the PRODUCT library accepts (csrc/attn_common.h, kProductVariantMask) and the tilings among bits 1-3
LEGACY_DECODE_PLAN, IN_LAUNCH_MERGE = 1 << 19, 1 << 20      # decode A/B selectors (csrc/attn_common.h); the second is lab-only
PRODUCT_VARIANT_MASK = (7 << 1) | (3 << 5) | (1 << 7) | (3 << 12) | LEGACY_DECODE_PLAN
PRODUCT_TILINGS = (0, 1, 4, 7)


def needs_lab(variant: int) -> bool:
    return bool(variant & ~PRODUCT_VARIANT_MASK) or ((variant >> 1) & 7) not in PRODUCT_TILINGS


def _bind(lib):
    lib.vattn_attn_workspace_bytes.restype = C.c_size_t
    lib.vattn_attn_workspace_bytes.argtypes = [C.POINTER(AttnParams)]
    lib.vattn_flash_attn_with_kvcache.restype = i32
    lib.vattn_flash_attn_with_kvcache.argtypes = [C.POINTER(AttnParams), vp]
    lib.vattn_hybrid_workspace_bytes.restype = C.c_size_t
    lib.vattn_hybrid_workspace_bytes.argtypes = [C.POINTER(AttnParams), C.POINTER(AttnParams)]
    lib.vattn_hybrid_attn.restype = i32
    lib.vattn_hybrid_attn.argtypes = [C.POINTER(AttnParams), C.POINTER(AttnParams), vp, vp]
    lib.vattn_cache_flat.restype = i32
    lib.vattn_cache_flat.argtypes = [vp, vp, vp, vp, i64, i32, i32, i64, i64, i64, i64, i32, vp]
    lib.vattn_cache_flat_rope.restype = i32
    lib.vattn_cache_flat_rope.argtypes = [vp, vp, vp, vp, i64, i32, i32, i64, i64, i64, i64, i32, vp, i64, i64, vp]
    lib.vattn_rotary_embedding.restype = i32
    lib.vattn_rotary_embedding.argtypes = [vp, vp, vp, i64, i32, i32, i32, i64, i64, i32, vp, i64, i32, i32, vp]
    lib.vattn_selftest_layouts.restype = i32
    lib.vattn_selftest_layouts.argtypes = [vp, C.POINTER(i32)]
    lib.vattn_time_attn.restype = C.c_float
    lib.vattn_time_attn.argtypes = [C.POINTER(AttnParams), vp, i32, i32]
    lib.vattn_kernels_last_error.restype = C.c_char_p
    lib.vattn_prefill_plan.restype = i32
    lib.vattn_prefill_plan.argtypes = [C.POINTER(AttnParams), C.POINTER(i32), C.POINTER(i32), C.POINTER(PrefillItem), i32, C.POINTER(PrefillItem), i32,
                                       C.POINTER(i32)]
    lib.vattn_prefill_plan_wg.restype = i32
    lib.vattn_prefill_plan_wg.argtypes = [C.POINTER(AttnParams), C.POINTER(i32), C.POINTER(i32), C.POINTER(PrefillItem), i32, C.POINTER(PrefillItem), i32,
                                          C.POINTER(i32), i32, C.POINTER(i32)]
    lib.vattn_attn_plan_describe.restype = i32
    lib.vattn_attn_plan_describe.argtypes = [C.POINTER(AttnParams), C.POINTER(PlanDesc)]
    lib.vattn_decode_plan.restype = i32
    lib.vattn_decode_plan.argtypes = [C.POINTER(AttnParams), C.POINTER(i32), C.POINTER(DecodeItem), i32, C.POINTER(i32)]
    return lib


def klib():
    """The product kernels (libvattn_amd.so)."""
    global _bound
    lib = L.lib()
    if not _bound:
        _bind(lib)
        _bound = True
    return lib


def klib_lab():
    """tools/lab/libvattn_lab.so: the same kernels built with -DVATTN_LAB (measurement scaffolding: alternative schedules, merge
    protocols, timing ablations).  Only reached through variants the product library rejects (tests, tools/kbench.py)."""
    global _lab
    if _lab is None:
        import os
        if os.environ.get("VATTN_NO_LAB") == "1":      # the product-only test run (tests/conftest.py, -m "gpu and not lab")
            raise RuntimeError("a lab kernel was requested in a product-only run (VATTN_NO_LAB=1): this test belongs to the `lab` marker")
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lab", "libvattn_lab.so")
        if not os.path.exists(path):
            raise RuntimeError("lab kernels requested (variant bits outside the product set) but tools/lab/libvattn_lab.so is missing: "
                               "run vattention_amd/build.py")
        _lab = _bind(C.CDLL(path))
    return _lab


def klib_for(variant: int):
    return klib_lab() if needs_lab(int(variant)) else klib()


def describe(p, lib=None) -> dict:
    """The launch plan of parameter block `p` (vattn_attn_plan_describe): pure host arithmetic."""
    d = PlanDesc()
    rc = (lib or klib()).vattn_attn_plan_describe(C.byref(p), C.byref(d))
    if rc != 0:
        raise RuntimeError(last_error(lib))
    return {n: int(getattr(d, n)) for n, _ in d._fields_}


def last_error(lib=None) -> str:
    return (lib or klib()).vattn_kernels_last_error().decode()


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def selftest_layouts(device=None):
    detail = (i32 * 8)()
    rc = klib().vattn_selftest_layouts(current_stream_ptr(device), detail)
    return rc, list(detail)


_DT = {torch.float16: 0, torch.bfloat16: 1}


def dtype_code(dt: torch.dtype) -> int:
    if dt not in _DT:
        raise RuntimeError("FlashAttention only support fp16 and bf16 data type")    # flash_api.cpp:1325-1326
    return _DT[dt]
