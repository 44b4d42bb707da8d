"""The `fa_vattn` attention backend: mirror of sarathi-lean's VAttentionFlashAttentionWrapper
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:17-224).

Same interface (init / begin_forward / set_batch_idx / forward / end_forward, is_profiling_iteration),
same per-iteration dataflow:
  prefill sequences first — append the chunk's K/V to the sequence's contiguous cache rows
  (cache_flat) and run causal attention of the chunk against the cache prefix [0, processed+chunk);
  then ONE batched decode call over the [:, :max_cache_len] view of every slot with the new K/V
  appended in-kernel at cache_seqlens and the slots selected by cache_batch_idx.
Both kernels are the gfx950 kernels of libvattn_amd.so.

Differences from the reference, on purpose:
  * slot indices are kept on the host as well, so selecting a prefill's row-block costs no
    device->host sync per prefill per layer (SURVEY §3.3 calls the reference's `.item()` there
    "a reference inefficiency worth not copying");
  * the decode call never hits the reference's transient "seqlen" error path, so
    decode rows are always computed.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .. import vattention as _vattention
from ..cache_ops import cache_flat, cache_flat_rope
from .. import flash_attn as _FA
from ..flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache
from .base_attention_wrapper import BaseAttentionWrapper
from . import timers as _T
from .timers import OperationMetrics


class VAttentionFlashAttentionWrapper(BaseAttentionWrapper):
    _inst = None

    def init(self, model_config, parallel_config, block_size: int, device: torch.device):
        super().init(model_config, parallel_config, block_size, device)
        self.is_metadata_initialized = False
        self.is_profiling_iteration = False
        self._rotary = None
        self._num_layers = model_config.get_num_layers(parallel_config)
        self._reset()

    def set_fused_rotary(self, cos_sin_cache: Optional[torch.Tensor]) -> None:
        """MI355X extension: hand the wrapper the model's rotary table ([max_position, rotary_dim],
        rotary_embedding.py:75-84) and pass UN-rotated q / k to forward(): RoPE is then applied inside the attention and
        cache-append launches (q and the new k rows in registers, the rotated k lands in the cache) instead of by a separate
        kernel before the wrapper (models/yi.py:172-173).  None switches back to the reference's dataflow."""
        self._rotary = cos_sin_cache

    def _reset(self):
        self.prefill_query_lens: List[int] = []
        self.prefill_cache_lens: List[int] = []
        self.current_total_len_device_lst: List[torch.Tensor] = []
        self.decode_cache_lens: Optional[torch.Tensor] = None
        self.batch_index: Optional[torch.Tensor] = None
        self.batch_index_gen: Optional[torch.Tensor] = None
        self._batch_index_host: Optional[List[int]] = None
        self.max_cache_len = 0
        self.decode_batch_size = 0
        self._decode_lens_host: List[int] = []
        self._dec_plan = None       # per-iteration launch plan of the decode call (built by layer 0, replayed by the other layers)
        self._pf_plans = {}         # per-iteration prefill work lists (flash_attn.prefill_plan), keyed by call site; same for every layer

    def get_cache_block(self, num_blocks: int, **kwargs):
        return None          # vAttention has no block tables

    def begin_forward(self, seq_metadata_list) -> None:
        self.is_profiling_iteration = False
        self.is_metadata_initialized = True
        self._dec_plan = None
        self._pf_plans = {}
        q_lens, c_lens, totals, dec = [], [], [], []
        for md in seq_metadata_list:
            if md.is_prompt:
                chunk = md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                done = md.seq.get_num_prompt_tokens_processed()
                q_lens.append(chunk)
                c_lens.append(done)
                totals.append(done + chunk)
        for md in seq_metadata_list:
            if not md.is_prompt:
                dec.append(md.seq.get_len() - 1)
        self.prefill_query_lens = q_lens
        self.prefill_cache_lens = c_lens
        # algorithmic work of this iteration's launches, per layer (only read by the op timers): prefill flops
        # 4.Hq.D.(n.c + n(n+1)/2) per chunk, decode bytes sum_b 2.len_b.Hkv.D.itemsize + q, o
        Hq, Hkv, D = self.num_q_heads, self.num_kv_heads, self.head_dim
        self._pf_flops = [4.0 * Hq * D * (n * c + n * (n + 1) / 2.0) for c, n in zip(c_lens, q_lens)]
        es = 2
        self._dc_bytes = float(sum(2 * (l + 1) * Hkv * D * es for l in dec) + len(dec) * Hq * D * es * 2)
        if totals:      # one H2D copy for all prefills, then views
            starts = [sum(q_lens[:i]) for i in range(len(q_lens))]
            meta = torch.tensor([totals, starts, q_lens], dtype=torch.int32, device=self.device)
            self.current_total_len_device_lst = [meta[0, i:i + 1] for i in range(len(totals))]
            self._prefill_totals, self._prefill_starts, self._prefill_qlens = meta[0], meta[1], meta[2]
        else:
            self.current_total_len_device_lst = []
        if not dec:
            self.decode_batch_size = 0
            return
        self.decode_batch_size = len(dec)
        self._decode_lens_host = dec
        self.decode_cache_lens = torch.tensor(dec, dtype=torch.int32, device=self.device)
        self.max_cache_len = max(dec) + 1

    def end_forward(self):
        self.is_metadata_initialized = False
        self._reset()

    def set_batch_idx(self, batch_idx: torch.Tensor, batch_idx_gen: torch.Tensor, batch_idx_host: Optional[List[int]] = None) -> None:
        self.batch_index = batch_idx.to(torch.int32)
        self.batch_index_gen = batch_idx_gen.to(torch.int32)
        # the cache engine passes the host copy it already has; a foreign caller costs one sync per iteration
        self._batch_index_host = list(batch_idx_host) if batch_idx_host is not None else [int(x) for x in batch_idx.tolist()]

    @staticmethod
    def _gate_layer(layer_id: Optional[int]) -> None:
        """Layer-ordered page mapping (vattention.enable_layered_async): this layer's pages must be mapped before its first kernel
        is launched.  A model that passes its layer index (llama.py:179-186) waits for that layer only; the reference's other models
        (yi.py, mistral.py, qwen.py, falcon.py, internlm.py) call forward() WITHOUT layer_id — then the first forward() of the
        iteration waits for every layer (nothing else is safe: the call could be any layer's)."""
        if layer_id is not None:
            _vattention.wait_layer(layer_id)
        else:
            _vattention.wait_all_layers()

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Tuple[torch.Tensor, torch.Tensor], softmax_scale: float = 1.0,
                layer_id: Optional[int] = None) -> torch.Tensor:
        assert self.is_metadata_initialized, "Metadata is not initialized."
        if self.is_profiling_iteration:
            return torch.zeros_like(query)       # memory-profiling pass: no attention (model_runner.py:192-201)
        self._gate_layer(layer_id)
        output = torch.empty_like(query)
        tok = self._forward_prefills(query, key, value, kv_cache, softmax_scale, layer_id, output)
        if self.decode_batch_size:
            self._forward_decodes(query, key, value, kv_cache, softmax_scale, layer_id, output, tok)
        return output

    def _forward_prefills(self, query, key, value, kv_cache, softmax_scale, layer_id, output, num_splits: int = 0) -> int:
        """Every prefill chunk of the iteration: cache_flat + causal attention against the cache prefix, written straight into
        the chunk's rows of `output`.  Returns the number of tokens consumed (the decode rows start there)."""
        Hq, Hkv, D = self.num_q_heads, self.num_kv_heads, self.head_dim
        k_all, v_all = kv_cache
        P = len(self.prefill_query_lens)
        if P >= 2 and max(self.prefill_query_lens) >= 2:
            # several prompts / chunks in one iteration (vLLM scheduler with short prompts): append each chunk's K/V, then ONE
            # batched launch over all of them instead of the reference's one attention call per prompt (:129-174)
            tok = 0
            with self.get_timer(OperationMetrics.ATTN_KV_CACHE_SAVE, layer_id):
                for i, (c_len, q_len) in enumerate(zip(self.prefill_cache_lens, self.prefill_query_lens)):
                    slot = self._batch_index_host[i]
                    if self._rotary is not None:
                        cache_flat_rope(key[tok:tok + q_len].view(q_len, Hkv, D), value[tok:tok + q_len].view(q_len, Hkv, D),
                                        k_all[slot][c_len:], v_all[slot][c_len:], self._rotary, c_len)
                    else:
                        cache_flat(key[tok:tok + q_len].view(q_len, Hkv, D), value[tok:tok + q_len].view(q_len, Hkv, D),
                                   k_all[slot][c_len:], v_all[slot][c_len:], "auto")
                    tok += q_len
            with self.get_timer(OperationMetrics.ATTN_PREFILL, layer_id) as tm:
                tm.work = sum(self._pf_flops)
                flash_attn_varlen_with_kvcache(query[:tok].view(tok, Hq, D), k_all, v_all, self._prefill_starts, self._prefill_qlens,
                                               max(self.prefill_query_lens), self._prefill_totals, self.batch_index[:P],
                                               softmax_scale=softmax_scale, causal=True, out=output[:tok].view(tok, Hq, D),
                                               num_splits=num_splits, _rotary_cos_sin=self._rotary,
                                               _max_seqlen_k=max(c + n for c, n in zip(self.prefill_cache_lens, self.prefill_query_lens)),
                                               _pf_plan=self._prefill_plan("varlen", self.prefill_query_lens,
                                                                           [c + n for c, n in zip(self.prefill_cache_lens, self.prefill_query_lens)], num_splits))
            return tok
        tok = 0
        for i, (c_len, q_len) in enumerate(zip(self.prefill_cache_lens, self.prefill_query_lens)):
            slot = self._batch_index_host[i]
            with self.get_timer(OperationMetrics.ATTN_INPUT_RESHAPE, layer_id):
                q = query[tok:tok + q_len].view(1, q_len, Hq, D)
                k = key[tok:tok + q_len].view(q_len, Hkv, D)
                v = value[tok:tok + q_len].view(q_len, Hkv, D)
                k_rows = k_all[slot]             # [max_ctx, kvh, D]: the slot's whole (virtual) row-block
                v_rows = v_all[slot]
            with self.get_timer(OperationMetrics.ATTN_KV_CACHE_SAVE, layer_id):
                if self._rotary is not None:
                    cache_flat_rope(k, v, k_rows[c_len:], v_rows[c_len:], self._rotary, c_len)
                else:
                    cache_flat(k, v, k_rows[c_len:], v_rows[c_len:], "auto")
            with self.get_timer(OperationMetrics.ATTN_PREFILL, layer_id) as tm:
                tm.work = self._pf_flops[i]
                # the kernel writes straight into this sequence's rows of `output` (no [q_len, Hq*D] copy afterwards)
                flash_attn_with_kvcache(q, k_rows.unsqueeze(0), v_rows.unsqueeze(0),
                                        cache_seqlens=self.current_total_len_device_lst[i],
                                        causal=True, softmax_scale=softmax_scale,
                                        out=output[tok:tok + q_len].view(1, q_len, Hq, D), _max_seqlen_k=c_len + q_len,
                                        num_splits=num_splits, _rotary_cos_sin=self._rotary,
                                        _pf_plan=self._prefill_plan(i, [q_len], [c_len + q_len], num_splits) if q_len > 1 else None)
            tok += q_len
        return tok

    def _prefill_plan(self, key, q_lens, k_lens, num_splits: int = 0):
        """The work list of one prefill call site of this iteration (flash_attn.prefill_plan: host arithmetic + one small H2D copy),
        built by the first layer that gets here and shared by the others — it depends on the lengths only.  None when the call cannot
        take a list anyway (an explicit split count, or the call is being recorded for the fused prefill || decode launch)."""
        if num_splits != 0 or _FA._capture_active():
            return None
        pl = self._pf_plans.get(key)
        if pl is None and self.head_dim == 128:
            import ctypes as C
            from .. import kernels as K
            p = K.AttnParams()
            p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = len(q_lens), max(q_lens), self.num_q_heads, self.num_kv_heads, self.head_dim, 1
            pl = self._pf_plans[key] = _FA.prefill_plan(p, q_lens, k_lens, self.device)
        return pl

    def _forward_decodes(self, query, key, value, kv_cache, softmax_scale, layer_id, output, tok: int) -> None:
        """ONE batched decode call: new K/V appended in-kernel at cache_seqlens of the slots named by cache_batch_idx.
        The L layers of an iteration issue the SAME call on different tensors: layer 0 builds the parameter block through the normal
        entry point (all argument checks), the other layers patch the six tensor pointers into it and launch — with small decode
        batches the launch is tens of microseconds and the Python argument handling would otherwise be the bottleneck."""
        Hq, Hkv, D = self.num_q_heads, self.num_kv_heads, self.head_dim
        k_all, v_all = kv_cache
        nb = self.decode_batch_size
        plan = self._dec_plan
        sig = (query.stride(0), key.stride(0), value.stride(0), output.stride(0), k_all.stride(), v_all.stride(), tok, float(softmax_scale),
               query.dtype, k_all.dtype)
        # op timers of a DECODE-ONLY iteration: one event pair around its L back-to-back launches (timers.group_begin) instead of one
        # pair per launch — an event costs microseconds of its own, which is 10-20 % of a batch-1 decode launch
        grouped = _T.op_timers_enabled() and not self.prefill_query_lens and layer_id is not None
        if grouped:
            if layer_id == 0:
                _T.group_begin("attn_decode")
            _T.group_add("attn_decode", self._dc_bytes)
        if plan is not None and plan["sig"] == sig and not _FA._capture_active():
            if grouped:
                _FA.relaunch(plan["p"], query.data_ptr() + plan["q_off"], key.data_ptr() + plan["k_off"], value.data_ptr() + plan["v_off"],
                             output.data_ptr() + plan["o_off"], k_all.data_ptr(), v_all.data_ptr(), self.device)
                if layer_id == self._num_layers - 1:
                    _T.group_end("attn_decode")
                return
            with self.get_timer(OperationMetrics.ATTN_DECODE, layer_id) as tm:
                tm.work = self._dc_bytes
                _FA.relaunch(plan["p"], query.data_ptr() + plan["q_off"], key.data_ptr() + plan["k_off"], value.data_ptr() + plan["v_off"],
                             output.data_ptr() + plan["o_off"], k_all.data_ptr(), v_all.data_ptr(), self.device)
            return
        capture = None if _FA._capture_active() else []
        with self.get_timer(OperationMetrics.ATTN_INPUT_RESHAPE, layer_id):
            dq = query[tok:tok + nb].view(nb, 1, Hq, D)
            dk = key[tok:tok + nb].view(nb, 1, Hkv, D)
            dv = value[tok:tok + nb].view(nb, 1, Hkv, D)
        import contextlib
        with (contextlib.nullcontext(_T.OpTimer("attn_decode")) if grouped else self.get_timer(OperationMetrics.ATTN_DECODE, layer_id)) as tm:
            tm.work = self._dc_bytes
            flash_attn_with_kvcache(dq, k_all[:, :self.max_cache_len], v_all[:, :self.max_cache_len], dk, dv,
                                    cache_seqlens=self.decode_cache_lens, block_table=None,
                                    softmax_scale=softmax_scale, causal=True,
                                    cache_batch_idx=self.batch_index_gen,
                                    out=output[tok:tok + nb].view(nb, 1, Hq, D), _rotary_cos_sin=self._rotary, _params_out=capture)
            # (no host-side lengths: the launch balances a ragged batch from `cache_seqlens` on the device, csrc/decode_body.h)
        if grouped and layer_id == self._num_layers - 1:
            _T.group_end("attn_decode")
        if capture:
            es = query.element_size()
            self._dec_plan = {"sig": sig, "p": capture[0], "q_off": tok * query.stride(0) * es, "k_off": tok * key.stride(0) * es,
                              "v_off": tok * value.stride(0) * es, "o_off": tok * output.stride(0) * es}
