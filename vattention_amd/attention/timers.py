"""Per-operation timers for the wrapper phases.  The reference wraps each phase in a CudaTimer keyed by
OperationMetrics (/root/reference/sarathi-lean/sarathi/metrics/cuda_timer.py:36-65,
vattention_flashattention_wrapper.py:134-218) that is inert unless op-level metrics are enabled; here the
timer records HIP events (on torch's current stream = the stream the kernels are launched on) only after
`enable_op_timers(True)`.

Two additions for bench.py's roofline objects:
  * sampling — `enable_op_timers(True, every=k)` times every k-th invocation of each operation and counts all of
    them, so a replay with half a million tiny launches is not slowed by a million events;
  * work — the wrapper attaches the ALGORITHMIC work of the timed launch (`timer.work = flops or bytes`), so the
    achieved rate of ragged launches is sum(work of the timed launches) / sum(their durations).
"""
from __future__ import annotations

import enum
from collections import defaultdict

import torch

_enabled = False
_every = 1
_records = defaultdict(list)       # name -> [(start event, end event, work)]
_counts = defaultdict(int)         # name -> invocations seen while enabled (timed or not)


class OperationMetrics(enum.Enum):
    ATTN_INPUT_RESHAPE = "attn_input_reshape"
    ATTN_KV_CACHE_SAVE = "attn_kv_cache_save"
    ATTN_PREFILL = "attn_prefill"
    ATTN_DECODE = "attn_decode"
    ATTN_OUTPUT_RESHAPE = "attn_output_reshape"


def enable_op_timers(on: bool, every: int = 1) -> None:
    global _enabled, _every
    _enabled = bool(on)
    _every = max(1, int(every))


def op_timers_enabled() -> bool:
    return _enabled


def drain_op_timers_detail() -> dict:
    """Synchronise and return {operation: {"ms": time of the TIMED invocations, "timed": how many were timed, "n": invocations
    seen, "work": algorithmic work of the timed ones (0 when the caller attached none)}}; clears the records."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in _records.items():
        out[name] = {"ms": sum(a.elapsed_time(b) for a, b, _ in evs), "timed": len(evs), "n": _counts.get(name, len(evs)),
                     "work": float(sum(w for _, _, w in evs))}
    _records.clear()
    _counts.clear()
    return out


def drain_op_timers() -> dict:
    """{operation name: estimated total milliseconds} (timed share scaled to all invocations); clears the records."""
    return {k: v["ms"] * (v["n"] / v["timed"] if v["timed"] else 0.0) for k, v in drain_op_timers_detail().items()}


class OpTimer:
    def __init__(self, operation, layer_id=None):
        self.name = operation.value if isinstance(operation, enum.Enum) else str(operation)
        self.layer_id = layer_id
        self.work = 0.0
        self._start = None

    def __enter__(self):
        if _enabled:
            n = _counts[self.name]
            _counts[self.name] = n + 1
            if n % _every == 0:
                self._start = torch.cuda.Event(enable_timing=True)
                self._start.record()
        return self

    def __exit__(self, *exc):
        if self._start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _records[self.name].append((self._start, end, self.work))
            self._start = None
        return False
