"""Hybrid-batch backend (`fa_streams`, also selected by `fa_pod`): prefill chunks and the decode batch of one iteration run
CONCURRENTLY on two HIP streams.

Mirrors the interface of the reference's VAttentionFlashAttentionStreamsWrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_streams_wrapper.py:17-237) and stands
in for its POD wrapper (vattention_flashattention_pod_wrapper.py:121-203): the reference fuses a prefill
and a decode kernel into one CUDA launch because two CUDA streams do not co-schedule well on its hardware; on MI355X the
command processor dispatches workgroups of both kernels as CUs free up, so the matrix-bound prefill chunk and the HBM-bound
decode batch overlap whenever the prefill grid leaves CUs (or register-file room) unused — measured 1.08-1.34x over the serial
order of two single-pass kernels on Sarathi-shaped hybrid batches (tools/hybrid_probe.py, profiles/r01_hybrid_probe.txt), 1.00x
when the prefill alone fills the chip, and 0.87-0.98x (slower) when a chip-filling prefill is forced to share.  The alternative for an underfilled prefill grid is to split its key range over the whole chip (KV-split,
DESIGN.md §5) and run decode after it; begin_forward() picks per iteration whichever a host-side estimate says is cheaper.

Differences from the reference's streams wrapper, on purpose: the join is a device-side event wait on the caller's stream (the
reference calls stream.synchronize() — a host stall — twice per layer, :187,235-236); outputs are written in place; the decode
stream is created non-blocking (hipMemMap must not wait for it: DESIGN.md §3).  Prefill and decode sequences are disjoint cache
slots and disjoint output rows, so the two streams share no data.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import vattention as _vattention

from .vattention_flashattention_wrapper import VAttentionFlashAttentionWrapper


class VAttentionFlashAttentionStreamsWrapper(VAttentionFlashAttentionWrapper):
    _inst = None
    # planning constants [measured, profiles/r01_kbench.txt]: sustained decode bandwidth, prefill rate of a full / one workgroup
    DECODE_BPS = 5.5e12
    PREFILL_FLOPS = 9.0e14
    CUS = 256

    def init(self, model_config, parallel_config, block_size: int, device: torch.device):
        super().init(model_config, parallel_config, block_size, device)
        self.decode_stream = torch.cuda.Stream(device=device)
        self._overlap = False

    def begin_forward(self, seq_metadata_list) -> None:
        super().begin_forward(seq_metadata_list)
        self._overlap = self._plan_overlap()

    def _plan_overlap(self) -> bool:
        """Two ways to run a hybrid iteration; pick the cheaper by a host-side estimate (same decision for every layer):
          overlap : decode on the side stream beside an UNSPLIT prefill (its grid leaves CUs free)   ~ max(t_prefill_unsplit, t_decode)
          serial  : prefill with its key range split over the whole chip, then decode                ~ t_prefill_split + t_decode
        (a KV-split prefill fills every CU, so overlapping it with decode only makes both slower: 0.87-0.90x measured)."""
        if not self.prefill_query_lens or not self.decode_batch_size:
            return False
        Hq, Hkv, D = self.num_q_heads, self.num_kv_heads, self.head_dim
        t_dec = sum(2.0 * (n + 1) * Hkv * D * 2 for n in self._decode_lens_host) / self.DECODE_BPS
        t_unsplit = t_split = 0.0
        for c, n in zip(self.prefill_cache_lens, self.prefill_query_lens):
            flops = 4.0 * Hq * D * (n * c + n * (n + 1) / 2.0)
            wg = -(-n // 256) * Hq                                   # 8-wave workgroups of this chunk
            if wg * 4 > self.CUS * 3:
                return False                                         # the chunk alone (nearly) fills the chip: nothing to overlap into
            rounds = -(-wg // self.CUS)
            per_wg = flops / wg
            t_unsplit += rounds * per_wg / (self.PREFILL_FLOPS / self.CUS)
            t_split += flops / self.PREFILL_FLOPS
        return max(t_unsplit, t_dec) < t_split + t_dec

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Tuple[torch.Tensor, torch.Tensor], softmax_scale: float = 1.0,
                layer_id: Optional[int] = None) -> torch.Tensor:
        assert self.is_metadata_initialized, "Metadata is not initialized."
        if self.is_profiling_iteration:
            return torch.zeros_like(query)
        if not self._overlap:                        # not a hybrid batch, or splitting the prefill beats overlapping it
            return super().forward(query, key, value, kv_cache, softmax_scale, layer_id)
        self._gate_layer(layer_id)
        output = torch.empty_like(query)
        main = torch.cuda.current_stream(self.device)
        side = self.decode_stream
        side.wait_stream(main)                       # q / k / v (and `output`) are produced on the caller's stream
        tok = sum(self.prefill_query_lens)
        with torch.cuda.stream(side):                # decode first: its workgroups start while the prefill is being enqueued
            self._forward_decodes(query, key, value, kv_cache, softmax_scale, layer_id, output, tok)
        self._forward_prefills(query, key, value, kv_cache, softmax_scale, layer_id, output, num_splits=1)
        main.wait_stream(side)                       # device-side join; no host synchronisation
        return output
