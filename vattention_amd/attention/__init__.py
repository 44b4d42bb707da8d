"""Attention-backend registry: mirror of /root/reference/sarathi-lean/sarathi/model_executor/attention/__init__.py:36-201.

Every backend name of the reference is known; the names that select the `fa_vattn` family resolve to
the MI355X-native wrapper.  Backends that exist in the reference only to wrap *other* CUDA libraries
(FlashInfer, FA3, paged baselines — SURVEY §2.1 row 6) raise
NotImplementedError by name instead of silently mapping to something else.
"""
from __future__ import annotations

from enum import Enum
from typing import Union

from .base_attention_wrapper import BaseAttentionWrapper  # noqa: F401
from .no_op_attention_wrapper import NoOpAttentionWrapper
from .vattention_flashattention_pod_wrapper import VAttentionFlashAttentionPodWrapper
from .vattention_flashattention_streams_wrapper import VAttentionFlashAttentionStreamsWrapper
from .vattention_flashattention_wrapper import VAttentionFlashAttentionWrapper


class AttentionBackend(Enum):
    FA_PAGED = "FA_PAGED"
    FI_PAGED = "FI_PAGED"
    FA_VATTN = "FA_VATTN"
    FI_VATTN = "FI_VATTN"
    FA_VATTN_SYNC = "FA_VATTN_SYNC"
    FI_VATTN_SYNC = "FI_VATTN_SYNC"
    FI_UNPAGED = "FI_UNPAGED"
    NO_OP = "NO_OP"
    FA3_VATTN = "FA3_VATTN"
    FA3_VATTN_SYNC = "FA3_VATTN_SYNC"
    FA_VATTN_MEGACACHE = "FA_VATTN_MEGACACHE"
    FA_VATTN_MEGACACHE_SYNC = "FA_VATTN_MEGACACHE_SYNC"
    FA_POD = "FA_POD"
    FA_STREAMS = "FA_STREAMS"
    FI_SERIAL_PAGED = "FI_SERIAL_PAGED"
    FA_POD_MEGACACHE = "FA_POD_MEGACACHE"
    FA_STREAMS_MEGACACHE = "FA_STREAMS_MEGACACHE"

    @staticmethod
    def _in(cfg, names):
        return str(cfg).upper() in names

    @staticmethod
    def is_vATTN(cfg) -> bool:
        return AttentionBackend._in(cfg, _VATTN)

    @staticmethod
    def is_attn_contiguous(cfg) -> bool:
        return AttentionBackend._in(cfg, _VATTN)

    @staticmethod
    def is_vATTN_SYNC(cfg) -> bool:
        return AttentionBackend._in(cfg, _VATTN_SYNC)

    @staticmethod
    def is_vLLM(cfg) -> bool:
        return AttentionBackend._in(cfg, _VLLM)


_VATTN = {"FA_VATTN", "FI_VATTN", "FA_VATTN_SYNC", "FI_VATTN_SYNC", "FA3_VATTN", "FA3_VATTN_SYNC", "FA_VATTN_MEGACACHE",
          "FA_VATTN_MEGACACHE_SYNC", "FA_POD", "FA_STREAMS", "FA_POD_MEGACACHE", "FA_STREAMS_MEGACACHE"}
_VATTN_SYNC = {"FA_VATTN_SYNC", "FI_VATTN_SYNC", "FA3_VATTN_SYNC", "FA_VATTN_MEGACACHE_SYNC"}
_VLLM = {"FA_PAGED", "FI_PAGED", "FI_UNPAGED", "FI_SERIAL_PAGED"}
_NATIVE = {AttentionBackend.FA_VATTN, AttentionBackend.FA_VATTN_SYNC, AttentionBackend.FA_VATTN_MEGACACHE,
           AttentionBackend.FA_VATTN_MEGACACHE_SYNC}

# hybrid-batch backends: prefill || decode on two HIP streams (the reference's FA_STREAMS) and as one fused launch (FA_POD)
_NATIVE_HYBRID = {AttentionBackend.FA_STREAMS, AttentionBackend.FA_STREAMS_MEGACACHE}
_NATIVE_POD = {AttentionBackend.FA_POD, AttentionBackend.FA_POD_MEGACACHE}

ATTENTION_BACKEND = AttentionBackend.NO_OP


def get_attn_type() -> str:
    return ATTENTION_BACKEND.value


def set_attention_backend(backend: Union[str, AttentionBackend]) -> None:
    global ATTENTION_BACKEND
    if isinstance(backend, str):
        name = backend.upper()
        if name not in AttentionBackend.__members__:
            raise ValueError(f"Unsupported attention backend: {backend}")
        backend = AttentionBackend[name]
    elif not isinstance(backend, AttentionBackend):
        raise ValueError(f"Unsupported attention backend: {backend}")
    ATTENTION_BACKEND = backend


def get_attention_wrapper():
    if ATTENTION_BACKEND == AttentionBackend.NO_OP:
        return NoOpAttentionWrapper.get_instance()
    if ATTENTION_BACKEND in _NATIVE:
        return VAttentionFlashAttentionWrapper.get_instance()
    if ATTENTION_BACKEND in _NATIVE_HYBRID:
        return VAttentionFlashAttentionStreamsWrapper.get_instance()
    if ATTENTION_BACKEND in _NATIVE_POD:
        return VAttentionFlashAttentionPodWrapper.get_instance()
    raise NotImplementedError(
        f"attention backend {ATTENTION_BACKEND.value} wraps a CUDA-only library in the reference and has no "
        "MI355X-native counterpart here; use FA_VATTN / FA_VATTN_SYNC / FA_VATTN_MEGACACHE[_SYNC] / FA_STREAMS / FA_POD[_MEGACACHE]")


def is_vattention_backend() -> bool:
    return ATTENTION_BACKEND.value in _VATTN


def is_vLLM_backend() -> bool:
    return ATTENTION_BACKEND.value in _VLLM


def is_attn_contiguous() -> bool:
    return ATTENTION_BACKEND.value in _VATTN
