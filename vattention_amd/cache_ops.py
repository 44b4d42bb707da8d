"""Drop-in for `sarathi.cache_ops.cache_flat` (/root/reference/sarathi-lean/csrc/cache.cpp:40-46,69-72;
kernel /root/reference/sarathi-lean/csrc/cache_kernels.cu:482-570): append n new tokens of K and V
([n, kvh, D]) to caller-sliced contiguous cache rows, in place, on the current stream."""
from __future__ import annotations

import torch

from . import kernels as K


def cache_flat(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
               kv_cache_dtype: str) -> None:
    if kv_cache_dtype != "auto":
        raise RuntimeError("Unsupported data type of kv cache: " + str(kv_cache_dtype))       # cache_kernels.cu:532-534
    if not (key.is_cuda and value.is_cuda and k_cache.is_cuda and v_cache.is_cuda):
        raise RuntimeError("vattention_amd.cache_ops: tensors must live on the GPU (there is no CPU path)")
    if k_cache.stride(0) != v_cache.stride(0):
        raise RuntimeError("k_cache.stride(0) == v_cache.stride(0)")                          # TORCH_CHECK at :543
    n, nh, hs = key.shape[0], key.shape[1], key.shape[2]
    if n == 0:
        return
    for t in (key, value, k_cache, v_cache):
        if t.stride(-1) != 1 or t.stride(-2) != hs:
            raise RuntimeError("cache_flat expects [tokens, heads, head_size] with contiguous (heads, head_size)")
    rc = K.klib().vattn_cache_flat(key.data_ptr(), value.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), n, nh, hs,
                                   key.stride(0), value.stride(0), k_cache.stride(0), v_cache.stride(0),
                                   key.element_size(), K.current_stream_ptr(key.device))
    if rc != 0:
        raise RuntimeError(K.last_error())


def cache_flat_rope(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    cos_sin_cache: torch.Tensor, pos0: int) -> None:
    """MI355X extension: cache_flat with the rotary embedding of the KEY rows fused in — k_cache[t] = rope(key[t]) at
    position pos0 + t, v_cache[t] = value[t] — one pass over the new K/V instead of the reference's rotary kernel followed by
    cache_flat (models/yi.py:172-173 + vattention_flashattention_wrapper.py:151-156).  `cos_sin_cache` is the model's
    [max_position, rotary_dim] table (rotary_embedding.py:75-84); NeoX pairing, rotary_dim == head size."""
    if not (key.is_cuda and value.is_cuda and k_cache.is_cuda and v_cache.is_cuda and cos_sin_cache.is_cuda):
        raise RuntimeError("vattention_amd.cache_ops: tensors must live on the GPU (there is no CPU path)")
    n, nh, hs = key.shape[0], key.shape[1], key.shape[2]
    if n == 0:
        return
    if cos_sin_cache.dtype != key.dtype or cos_sin_cache.shape[1] != hs or cos_sin_cache.stride(1) != 1:
        raise RuntimeError("cos_sin_cache must be [positions, head_size] in the key's dtype")
    if pos0 + n > cos_sin_cache.shape[0]:
        raise RuntimeError("positions exceed the cos/sin table")
    for t in (key, value, k_cache, v_cache):
        if t.stride(-1) != 1 or t.stride(-2) != hs:
            raise RuntimeError("cache_flat_rope expects [tokens, heads, head_size] with contiguous (heads, head_size)")
    rc = K.klib().vattn_cache_flat_rope(key.data_ptr(), value.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), n, nh, hs,
                                        key.stride(0), value.stride(0), k_cache.stride(0), v_cache.stride(0), K.dtype_code(key.dtype),
                                        cos_sin_cache.data_ptr(), cos_sin_cache.stride(0), int(pos0), K.current_stream_ptr(key.device))
    if rc != 0:
        raise RuntimeError(K.last_error())


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """Drop-in for sarathi's pos_encoding_ops.rotary_embedding (csrc/pos_encoding_kernels.cu:80-129): in place on query
    [T, Hq*hs] and key [T, Hkv*hs].  The UNFUSED path — what the fused launches are measured against."""
    if not (query.is_cuda and key.is_cuda and positions.is_cuda and cos_sin_cache.is_cuda):
        raise RuntimeError("vattention_amd.cache_ops: tensors must live on the GPU (there is no CPU path)")
    if positions.dtype != torch.int64:
        raise RuntimeError("positions must be int64")
    T = query.shape[0]
    rc = K.klib().vattn_rotary_embedding(positions.data_ptr(), query.data_ptr(), key.data_ptr(), T, query.shape[1] // head_size,
                                         key.shape[1] // head_size, head_size, query.stride(0), key.stride(0), K.dtype_code(query.dtype),
                                         cos_sin_cache.data_ptr(), cos_sin_cache.stride(0), cos_sin_cache.shape[1], 1 if is_neox else 0,
                                         K.current_stream_ptr(query.device))
    if rc != 0:
        raise RuntimeError(K.last_error())
