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
