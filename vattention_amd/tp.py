"""Tensor-parallel glue for the hot path.

The KV cache shards by KV head with zero exchange (/root/reference/sarathi-lean/sarathi/config.py:139-167): every TP rank is
its own process with its own allocator and `num_kv_heads / TP` heads; `tokens_per_page` depends only on the per-rank head count,
so all ranks take identical page decisions from identical `seq_lens`.  The only cross-rank step on this path is the engine's
control-plane `min(free_blocks)` over workers (/root/reference/sarathi-lean/sarathi/engine/base_llm_engine.py:381-390), provided
here as a `torch.distributed` all-reduce (RCCL over xGMI when the backend is "nccl" on ROCm; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def heads_for_rank(num_q_heads: int, num_kv_heads: int, tp_size: int) -> Tuple[int, int]:
    """Per-rank (q heads, kv heads): config.py:139-167 — KV heads are divided, never below one (replicated for MQA)."""
    assert num_q_heads % tp_size == 0
    return num_q_heads // tp_size, max(1, num_kv_heads // tp_size)


def min_free_kvblocks(local_free: int, group=None, device=None) -> int:
    """min over ranks of num_free_kvblocks() — what the scheduler admits against."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(local_free)
    t = torch.tensor([min(int(local_free), (1 << 62))], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())
