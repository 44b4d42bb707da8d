"""Mirror of the reference's NoOpAttentionWrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/no_op_attention_wrapper.py:36-45):
the only non-computing backend; returns an uninitialised tensor shaped like the query."""
from __future__ import annotations

import torch

from .base_attention_wrapper import BaseAttentionWrapper


class NoOpAttentionWrapper(BaseAttentionWrapper):
    _inst = None

    def init(self, model_config, parallel_config, block_size, device):
        self.device = device

    def get_cache_block(self, num_blocks: int, **kwargs):
        return None

    def begin_forward(self, seq_metadata_list) -> None:
        pass

    def end_forward(self):
        pass

    def forward(self, query, key, value, kv_cache, softmax_scale: float = 1.0, layer_id=None):
        return torch.empty_like(query)
