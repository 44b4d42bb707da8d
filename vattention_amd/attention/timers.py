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
        out[name] = {"ms": sum(r[0].elapsed_time(r[1]) for r in evs), "timed": sum(r[3] if len(r) > 3 else 1 for r in evs),
                     "n": _counts.get(name, len(evs)), "work": float(sum(r[2] for r in evs))}
    _records.clear()
    _counts.clear()
    _groups.clear()
    return out


def drain_op_timers() -> dict:
    """{operation name: estimated total milliseconds} (timed share scaled to all invocations); clears the records."""
    return {k: v["ms"] * (v["n"] / v["timed"] if v["timed"] else 0.0) for k, v in drain_op_timers_detail().items()}


_groups = {}                       # name -> (start event, launches so far, work so far) of an open group


def group_begin(name: str) -> bool:
    """Open a GROUP measurement: ONE event pair around a run of back-to-back launches of the same operation (the L decode launches of a
    decode-only iteration).  A HIP event is a barrier packet with a signal — a few microseconds of its own — so bracketing every launch
    of a 30-60 us kernel with its own pair inflates each reading by 10-20 % (batch-1 decode at 128 k: 63 us per launch by per-launch
    events, 52 us by tools/kbench.py's pair around twenty launches); the group pays that once per L launches, and still contains every
    real gap between them.  Returns False (nothing opened) when timers are off or this iteration is not sampled."""
    if not _enabled:
        return False
    n = _counts[name + "#groups"]
    _counts[name + "#groups"] = n + 1
    if n % _every != 0:
        return False
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    _groups[name] = [ev, 0, 0.0]
    return True


def group_add(name: str, work: float) -> None:
    """Account one launch of a grouped iteration (the caller uses no timer of its own); if the iteration's group is open — it is one of
    the sampled ones — the launch is timed by the group's event pair."""
    _counts[name] += 1
    g = _groups.get(name)
    if g is not None:
        g[1] += 1
        g[2] += work


def group_end(name: str) -> None:
    g = _groups.pop(name, None)
    if g is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    _records[name].append((g[0], end, g[2], g[1]))


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
