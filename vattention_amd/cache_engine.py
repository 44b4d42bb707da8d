"""Mirror of sarathi-lean's vATTNCacheEngine and its registry
(/root/reference/sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py:25-191,
base_cache_engine.py:19-67, __init__.py:9-25): owns the seq_id -> slot table and the per-slot
current lengths, calls the `vattention` module once per iteration, and pushes the slot indices of
the iteration's prefills+decodes into the attention wrapper.

The config objects are duck-typed (only the attributes / methods the reference reads are used), so
sarathi-lean's own CacheConfig / ModelConfig / ParallelConfig work unchanged.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import os

import torch

from . import vattention
from .attention import AttentionBackend, get_attention_wrapper


def get_cache_engine(attn_backend: str):
    if AttentionBackend.is_vATTN(attn_backend):
        return vATTNCacheEngine
    raise NotImplementedError(f"Cache engine for {attn_backend} is not implemented (paged baselines are out of scope).")


def get_cache_mem_alloc_backend(attn_backend: str) -> str:
    if AttentionBackend.is_vATTN_SYNC(attn_backend):
        return "sync"
    if AttentionBackend.is_vATTN(attn_backend):
        return "async"
    return "noop"


class vATTNCacheEngine:
    def __init__(self, cache_config, model_config, parallel_config, mem_alloc_backend: str) -> None:
        self.cache_config, self.model_config, self.parallel_config = cache_config, model_config, parallel_config
        self.max_batch_size = cache_config.max_batch_size
        self.device = torch.empty(1).cuda().device
        self.device_idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.max_model_seq_len = model_config.max_model_len
        self.curr_seq_lens: List[int] = [0] * self.max_batch_size
        self.seq_to_batch_idx: Dict[int, int] = {}
        self.page_size = cache_config.page_size
        self.vattn_async = mem_alloc_backend == "async"
        self.vattn_mega_cache = "megacache" in str(model_config.attention_backend).lower()
        self.cache_mem_size = cache_config.memory_for_gpu
        self.head_size = model_config.get_head_size()
        self.num_layers = model_config.get_num_layers(parallel_config)
        self.num_heads = model_config.get_num_kv_heads(parallel_config)
        self.dtype = model_config.dtype
        self.layout_policy = self._apply_layout_policy(cache_config)
        self.block_size = getattr(cache_config, "block_size", None)
        self.num_gpu_blocks = getattr(cache_config, "num_gpu_blocks", None)
        self.curr_batch_idx = None
        self.gpu_cache = self.allocate_gpu_cache()

    def _apply_layout_policy(self, cache_config) -> str:
        """vattention_amd/policy.py: pools that would need > 100 k physical handles move to megacache + 8 MiB pages.  The layout is
        internal to the engine (the wrapper gets per-layer (K, V) views either way)."""
        from .policy import choose_layout
        keep = bool(getattr(cache_config, "vattn_keep_layout", False)) or os.environ.get("VATTN_KEEP_LAYOUT", "0") == "1"
        page, mega, what = choose_layout(self.page_size, self.vattn_mega_cache, self.cache_mem_size, keep)
        if what != "configured":
            import sys
            print("[vattn] layout policy -> " + what + "; set cache_config.vattn_keep_layout / VATTN_KEEP_LAYOUT=1 to keep the configured layout",
                  file=sys.stderr)
        self.page_size, self.vattn_mega_cache = page, mega
        return what

    def num_free_blocks(self) -> int:
        return vattention.num_free_kvblocks()

    def allocate_gpu_cache(self) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        # layer-ordered mapping of new prompts needs the wrapper to gate each layer (wait_layer): this package's wrappers do
        vattention.enable_layered_async(self.vattn_async and not self.vattn_mega_cache and
                                        os.environ.get("VATTN_LAYERED_ASYNC", "1") != "0")
        kv = vattention.init_kvcache(self.num_layers, self.num_heads, self.head_size, self.max_batch_size,
                                     self.max_model_seq_len, self.device_idx, self.dtype, self.page_size,
                                     self.vattn_mega_cache)
        if self.vattn_mega_cache:
            k, v = kv[0], kv[1]
            cache = [(k[:, :, l], v[:, :, l]) for l in range(self.num_layers)]
        else:
            cache = list(zip(kv[:self.num_layers], kv[self.num_layers:]))
        for k_l, v_l in cache:
            assert k_l.device == self.device and v_l.device == self.device
        vattention.reserve_physical_pages(self.cache_mem_size)
        return cache

    def preempt_requests(self, preempted_seq) -> None:
        for seq in preempted_seq:
            self.free_request(seq.seq_id)

    def get_k_cache(self, layer_idx: int) -> torch.Tensor:
        return self.gpu_cache[layer_idx][0]

    def get_v_cache(self, layer_idx: int) -> torch.Tensor:
        return self.gpu_cache[layer_idx][1]

    def step(self, seq_metadata_list) -> None:
        idx_prompt: List[int] = []
        idx_gen: List[int] = []
        for md in seq_metadata_list:
            seq = md.seq
            if md.is_prompt:
                ctx = seq.get_num_prompt_tokens_processed() + seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                slot = self.get_req_batch_idx(seq.seq_id, ctx)
                self.curr_seq_lens[slot] = ctx
                idx_prompt.append(slot)
            else:
                ctx = seq.get_len()
                slot = self.get_req_batch_idx(seq.seq_id, ctx)
                self.curr_seq_lens[slot] = ctx
                idx_gen.append(slot)
        if self.vattn_async:
            vattention.step_async(self.curr_seq_lens)
        else:
            vattention.step(self.curr_seq_lens, True)
        both = idx_prompt + idx_gen
        allidx = torch.tensor(both + idx_gen, dtype=torch.int32, device=self.device)    # ONE H2D copy
        self.curr_batch_idx = allidx[:len(both)]
        get_attention_wrapper().set_batch_idx(self.curr_batch_idx, allidx[len(both):], both)

    def prefetch_request(self, seq_id: int, seq_len: int) -> int:
        """MI355X extension: call once the scheduler knows which request it admits NEXT (before or while the current iteration runs):
        its slot is reserved and its pages are mapped by the mapper thread under the current forward pass, so that the iteration
        that starts the request maps nothing synchronously.  Returns the slot (-1: none free / already placed).  A request that is
        dropped before it starts is released like any other: free_request(seq_id) (its pages stay mapped and reclaimable)."""
        if seq_id in self.seq_to_batch_idx:
            return -1
        slot = vattention.premap(seq_len)
        if slot >= 0:
            self.seq_to_batch_idx[seq_id] = slot
        return slot

    def on_step_completion(self, seq_metadata_list) -> None:
        for md in seq_metadata_list:
            if md.seq.is_finished():
                self.free_request(md.seq.seq_id)

    def get_req_batch_idx(self, seq_id: int, seq_len: int) -> int:
        slot = self.seq_to_batch_idx.get(seq_id)
        return slot if slot is not None else self.alloc_new_batch_idx(seq_id, seq_len)

    def alloc_new_batch_idx(self, seq_id: int, seq_len: int) -> int:
        slot = vattention.alloc_new_batch_idx(seq_len)
        assert slot != -1, "Failed to allocate new batch idx. This is not expected..."
        self.seq_to_batch_idx[seq_id] = slot
        return slot

    def free_request(self, seq_id: int) -> None:
        slot = self.seq_to_batch_idx.pop(seq_id, None)
        if slot is None:
            raise Exception(f"seq_id {seq_id} not found in req_table")
        vattention.free_batch_idx(slot)
        self.curr_seq_lens[slot] = 0

    def reclaim_req_ids(self) -> None:
        for seq_id in list(self.seq_to_batch_idx):
            self.free_request(seq_id)

    def get_batch_idx(self) -> torch.Tensor:
        return self.curr_batch_idx

    def clear_batch_index(self) -> None:
        self.curr_batch_idx = None

    def release_kvcache_physical(self):
        vattention.release_kvcache_physical()

    def disable_deferred_reclamation(self):
        vattention.set_deferred_reclamation(False)

    @staticmethod
    def get_cache_block_size(block_size: int, model_config, parallel_config) -> int:
        head_size = model_config.get_head_size()
        num_heads = model_config.get_num_kv_heads(parallel_config)
        num_layers = model_config.get_num_layers(parallel_config)
        itemsize = torch.tensor([], dtype=model_config.dtype).element_size()
        return itemsize * num_layers * 2 * block_size * num_heads * head_size

    def cleanup_kvcache(self):
        vattention.cleanup()
