"""Per-operation timers for the wrapper phases.  The reference wraps each phase in a CudaTimer keyed by
OperationMetrics (/root/reference/sarathi-lean/sarathi/metrics/cuda_timer.py:36-65,
vattention_flashattention_wrapper.py:134-218) that is inert unless op-level metrics are enabled; here the
timer records HIP events only after `enable_op_timers(True)`."""
from __future__ import annotations

import enum
from collections import defaultdict

import torch

_enabled = False
_records = defaultdict(list)


class OperationMetrics(enum.Enum):
    ATTN_INPUT_RESHAPE = "attn_input_reshape"
    ATTN_KV_CACHE_SAVE = "attn_kv_cache_save"
    ATTN_PREFILL = "attn_prefill"
    ATTN_DECODE = "attn_decode"
    ATTN_OUTPUT_RESHAPE = "attn_output_reshape"


def enable_op_timers(on: bool) -> None:
    global _enabled
    _enabled = bool(on)


def drain_op_timers() -> dict:
    """Synchronise and return {operation name: total milliseconds}; clears the records."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in _records.items():
        out[name] = sum(a.elapsed_time(b) for a, b in evs)
    _records.clear()
    return out


class OpTimer:
    def __init__(self, operation, layer_id=None):
        self.name = operation.value if isinstance(operation, enum.Enum) else str(operation)
        self.layer_id = layer_id
        self._start = None

    def __enter__(self):
        if _enabled:
            self._start = torch.cuda.Event(enable_timing=True)
            self._start.record()
        return self

    def __exit__(self, *exc):
        if _enabled and self._start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _records[self.name].append((self._start, end))
            self._start = None
        return False
