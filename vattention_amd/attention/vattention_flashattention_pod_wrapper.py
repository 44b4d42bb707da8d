"""Hybrid-batch backend `fa_pod`: the prefill chunk(s) and the decode batch of one iteration go to the GPU as ONE fused launch
(vattn_hybrid_attn, vattention_amd/csrc/hybrid_kernels.hip) — the MI355X counterpart of the reference's POD-Attention wrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_pod_wrapper.py:121-203, which calls
pod_attn.true_fused_attn_with_kvcache once per layer for a [prefill chunk | decode batch] iteration).

Same metadata handling and per-part semantics as the base wrapper (cache_flat of the chunk's K/V first, then attention against the
cache prefix; the decode part appends its one row per sequence in-kernel).  The reference's wrapper slices the prefill part's
output without the `[prefill_cache_len:]` offset its non-fused sibling applies (:154-158); here both parts write straight into
their rows of `output`, as in the base wrapper.

The fused launch co-locates one matrix-bound and one HBM-bound workgroup on every CU.  MEASURED on MI355X
(profiles/r02_hybrid_probe.txt, r02_hybrid_e2e.txt) it LOSES to the serial order and to two streams on every Sarathi-shaped batch
(0.26-0.64x): the stand-alone kernels already saturate what each is bound by (HBM for decode at 12 waves per CU, board power for
prefill), and inside one kernel both bodies share one register allocation (256 per lane), which leaves decode a third of its
waves.  Round 4 closed the row: the fused launch is LAB-ONLY (built into tools/lab/libvattn_lab.so, parity-checked against the oracle by
tests/test_gpu_hybrid_fused.py); there is no environment switch any more.  `fa_pod` IS the streams wrapper's per-iteration policy;
the class attribute FUSED_ENABLED exists for the tests and tools/hybrid_*.py, which set it to drive the lab kernel end to end."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import vattention as _vattention
from ..flash_attn import hybrid_attn
from .vattention_flashattention_streams_wrapper import VAttentionFlashAttentionStreamsWrapper


class VAttentionFlashAttentionPodWrapper(VAttentionFlashAttentionStreamsWrapper):
    _inst = None
    FUSED_ENABLED = False         # tests / tools only (the lab library's fused launch)
    FUSE_MIN_SHARE = 0.15
    # stand-alone rates of the two bodies inside the fused launch [measured, profiles/r02_hybrid_probe.txt]
    FUSED_PREFILL_FLOPS = 4.5e14

    def begin_forward(self, seq_metadata_list) -> None:
        super().begin_forward(seq_metadata_list)
        self._fused = self._plan_fused()

    def _plan_fused(self) -> bool:
        if not self.FUSED_ENABLED:
            return False
        if not self.prefill_query_lens or not self.decode_batch_size or self.head_dim != 128:
            return False
        if len(self.prefill_query_lens) >= 2 and max(self.prefill_query_lens) < 2:
            return False
        Hq, Hkv, D = self.num_q_heads, self.num_kv_heads, self.head_dim
        t_dec = sum(2.0 * (n + 1) * Hkv * D * 2 for n in self._decode_lens_host) / self.DECODE_BPS
        flops = sum(4.0 * Hq * D * (n * c + n * (n + 1) / 2.0) for c, n in zip(self.prefill_cache_lens, self.prefill_query_lens))
        t_pre = flops / self.FUSED_PREFILL_FLOPS
        lo, hi = min(t_pre, t_dec), max(t_pre, t_dec)
        return lo >= self.FUSE_MIN_SHARE * hi

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Tuple[torch.Tensor, torch.Tensor], softmax_scale: float = 1.0,
                layer_id: Optional[int] = None) -> torch.Tensor:
        assert self.is_metadata_initialized, "Metadata is not initialized."
        if self.is_profiling_iteration:
            return torch.zeros_like(query)
        if not getattr(self, "_fused", False):
            return super().forward(query, key, value, kv_cache, softmax_scale, layer_id)
        self._gate_layer(layer_id)
        output = torch.empty_like(query)
        tok = sum(self.prefill_query_lens)
        # the chunk's cache_flat launches run normally (they precede the fused launch on the stream); the two attention calls are
        # recorded and fused
        hybrid_attn(lambda: self._forward_prefills(query, key, value, kv_cache, softmax_scale, layer_id, output, num_splits=1),
                    lambda: self._forward_decodes(query, key, value, kv_cache, softmax_scale, layer_id, output, tok),
                    self.device)
        return output
