"""Mirror of sarathi-lean's BaseAttentionWrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/base_attention_wrapper.py:12-68):
one process-wide wrapper instance per backend; `init` reads the per-GPU head counts from the model /
parallel config; `get_timer(op, layer)` hands out per-(operation, layer) timers."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Optional, Tuple, Union

import torch

from .timers import OperationMetrics, OpTimer  # noqa: F401


class BaseAttentionWrapper(ABC):
    _inst = None

    def init(self, model_config, parallel_config, block_size: int, device: torch.device):
        self.device = device
        self.num_q_heads = model_config.get_num_q_heads(parallel_config)
        self.num_kv_heads = model_config.get_num_kv_heads(parallel_config)
        self.head_dim = model_config.get_head_size()
        self.dtype = model_config.dtype
        self.block_size = block_size
        self._timers = {}

    def get_timer(self, operation, layer_id: Optional[int] = None):
        key = (operation, layer_id)
        t = self._timers.get(key)
        if t is None:
            t = self._timers[key] = OpTimer(operation, layer_id)
        return t

    @classmethod
    def get_instance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    @abstractmethod
    def begin_forward(self, seq_metadata_list) -> None:
        ...

    @abstractmethod
    def end_forward(self):
        ...

    @abstractmethod
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
                softmax_scale: float = 1.0, layer_id: Optional[int] = None) -> torch.Tensor:
        ...
