"""Hybrid-batch backend (`fa_streams`, also selected by `fa_pod`): prefill chunks and the decode batch of one iteration run
CONCURRENTLY on two HIP streams.

Mirrors the interface of the reference's VAttentionFlashAttentionStreamsWrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_streams_wrapper.py:17-237) and stands
in for its POD wrapper (vattention_flashattention_pod_wrapper.py:121-203): the reference fuses a prefill
and a decode kernel into one CUDA launch because two CUDA streams do not co-schedule well on its hardware; on MI355X the
command processor dispatches workgroups of both kernels as CUs free up, so the matrix-bound prefill chunk and the HBM-bound
decode batch overlap whenever the prefill grid leaves CUs (or register-file room) unused — measured 1.08-1.34x over the serial
order on Sarathi-shaped hybrid batches (tools/hybrid_probe.py, profiles/r01_hybrid_probe.txt), 1.00x when the prefill alone
fills the chip.

Differences from the reference's streams wrapper, on purpose: the join is a device-side event wait on the caller's stream (the
reference calls stream.synchronize() — a host stall — twice per layer, :187,235-236); outputs are written in place; the decode
stream is created non-blocking (hipMemMap must not wait for it: DESIGN.md §3).  Prefill and decode sequences are disjoint cache
slots and disjoint output rows, so the two streams share no data.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .vattention_flashattention_wrapper import VAttentionFlashAttentionWrapper


class VAttentionFlashAttentionStreamsWrapper(VAttentionFlashAttentionWrapper):
    _inst = None

    def init(self, model_config, parallel_config, block_size: int, device: torch.device):
        super().init(model_config, parallel_config, block_size, device)
        self.decode_stream = torch.cuda.Stream(device=device)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Tuple[torch.Tensor, torch.Tensor], softmax_scale: float = 1.0,
                layer_id: Optional[int] = None) -> torch.Tensor:
        assert self.is_metadata_initialized, "Metadata is not initialized."
        if self.is_profiling_iteration:
            return torch.zeros_like(query)
        if not self.prefill_query_lens or not self.decode_batch_size:      # not a hybrid batch: nothing to overlap
            return super().forward(query, key, value, kv_cache, softmax_scale, layer_id)
        output = torch.empty_like(query)
        main = torch.cuda.current_stream(self.device)
        side = self.decode_stream
        side.wait_stream(main)                       # q / k / v (and `output`) are produced on the caller's stream
        tok = sum(self.prefill_query_lens)
        with torch.cuda.stream(side):                # decode first: its workgroups start while the prefill is being enqueued
            self._forward_decodes(query, key, value, kv_cache, softmax_scale, layer_id, output, tok)
        self._forward_prefills(query, key, value, kv_cache, softmax_scale, layer_id, output)
        main.wait_stream(side)                       # device-side join; no host synchronisation
        return output
