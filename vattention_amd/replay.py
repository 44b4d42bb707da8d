"""Ray-free replay harness for the hot path: drives vATTNCacheEngine + the fa_vattn wrapper exactly the way
sarathi-lean's worker does (BaseWorker.execute_model -> cache_engine.step -> per-layer wrapper.forward ->
on_step_completion; /root/reference/sarathi-lean/sarathi/worker/base_worker.py:173-208,
model_executor/model_runner.py:227-259) with synthetic q/k/v in place of the transformer body.

The light Sequence / SequenceMetadata / *Config classes expose only what the wrapper and cache engine read
(/root/reference/sarathi-lean/sarathi/core/datatypes/sequence.py, sarathi/config.py:139-167).
Workloads restate the reference's run scripts:
  static trace  scripts/benchmark_e2e_static_trace.py:6-57 — all requests at t=0, equal lengths, P:D ratio,
                vLLM scheduler: whole prompts, prefills prioritised, decode batches up to max_batch_size.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

# per-GPU shapes: (num_layers, num_q_heads, num_kv_heads, head_size)  SURVEY §8
MODELS = {
    "yi-6b": (32, 32, 4, 128),
    "llama-3-8b": (32, 32, 8, 128),
    "yi-34b": (60, 56, 8, 128),
    "llama-3-70b": (80, 64, 8, 128),
    "falcon-7b": (32, 71, 1, 64),          # multi-query, head size 64
    "mistral-7b": (32, 32, 8, 128),
    "qwen-72b": (80, 64, 64, 128),         # MHA
}


@dataclass
class ParallelConfig:
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1


@dataclass
class ModelConfig:
    name: str
    dtype: torch.dtype = torch.float16
    max_model_len: int = 32768
    attention_backend: str = "fa_vattn"
    num_layers: int = 32
    num_q_heads: int = 32
    num_kv_heads: int = 4
    head_size: int = 128

    @classmethod
    def named(cls, name: str, **kw):
        L, hq, hkv, d = MODELS[name]
        return cls(name=name, num_layers=L, num_q_heads=hq, num_kv_heads=hkv, head_size=d, **kw)

    def get_num_q_heads(self, pc):
        return self.num_q_heads // pc.tensor_parallel_size

    def get_num_kv_heads(self, pc):        # config.py:139-167: KV heads divided per TP rank (at least 1)
        return max(1, self.num_kv_heads // pc.tensor_parallel_size)

    def get_head_size(self):
        return self.head_size

    def get_num_layers(self, pc):
        return self.num_layers // pc.pipeline_parallel_size


@dataclass
class CacheConfig:
    page_size: int
    max_batch_size: int
    memory_for_gpu: int
    block_size: Optional[int] = None
    num_gpu_blocks: Optional[int] = None
    vattn_keep_layout: bool = False      # True: never let the cache engine's layout policy replace the configured page size / layout


class Sequence:
    def __init__(self, seq_id: int, prompt_len: int, total_len: int):
        self.seq_id, self.prompt_len, self.total_len = seq_id, prompt_len, total_len
        self.prompt_processed = 0
        self.output_len = 0

    def get_next_prompt_chunk_len(self, chunk_size: int) -> int:
        return min(chunk_size, self.prompt_len - self.prompt_processed)

    def get_num_prompt_tokens_processed(self) -> int:
        return self.prompt_processed

    def get_len(self) -> int:
        return self.prompt_len + self.output_len

    def is_finished(self) -> bool:
        return self.get_len() >= self.total_len

    @property
    def prompt_done(self) -> bool:
        return self.prompt_processed >= self.prompt_len


@dataclass
class SequenceMetadata:
    seq: Sequence
    prompt_chunk_len: int
    is_prompt: bool


@dataclass
class ReplayStats:
    iterations: int = 0
    prefill_tokens: int = 0
    decode_tokens: int = 0
    kv_util_samples: List[float] = field(default_factory=list)     # live-token bytes / mapped bytes
    mapped_over_reserved: List[float] = field(default_factory=list)
    kv_needed_samples: List[tuple] = field(default_factory=list)   # (live / needed-page tokens, active slots)
    # attention work issued, per layer and per query head: (query row, visible key) pairs of the prefill form, keys read by the decode
    # form — what bench.py's CPU baseline scales its measured pairs/s with
    prefill_pairs: float = 0.0
    decode_pairs: float = 0.0
    iter_events: List[object] = field(default_factory=list)         # one HIP event per iteration end (time_iterations)
    iter_phase: List[int] = field(default_factory=list)             # 0 = requests still waiting (steady state), 1 = drain tail
    iter_util: List[tuple] = field(default_factory=list)            # per iteration: (live/mapped, live/needed, active slots, mapped/pool)
    # external fragmentation, sampled every iteration: free pool pages that NO request can use / all physical pages.  A page-group is
    # one page in each of the 2L tensors (megacache: 2), any free page serves any slot and any position, so the only unusable free
    # pages are those left over when the pool holds no whole group (pool_pages mod pages_per_group)
    ext_frag_max: float = 0.0
    ext_frag_samples: int = 0


class HotPathRunner:
    """Owns the cache engine + wrapper for one GPU and replays scheduler iterations."""

    def __init__(self, model: ModelConfig, parallel: ParallelConfig, cache: CacheConfig, device="cuda:0", seed=42, reference=None):
        """reference: None = this package's wrapper + cache engine; else an object with `.wrapper` (an attention-wrapper INSTANCE) and
        `.engine_cls` — tools/ref_wrapper_bench.py passes the REFERENCE's own classes (tests/ref_loader.py), which only know the
        reference's API: no admission look-ahead, no layer-ordered mapping, no host-side hints."""
        from .attention import get_attention_wrapper, set_attention_backend
        from .cache_engine import get_cache_engine, get_cache_mem_alloc_backend
        self.model, self.parallel, self.cache_cfg = model, parallel, cache
        self.device = torch.device(device)
        if reference is not None:
            from . import vattention as _va0
            _va0.enable_layered_async(False)         # the reference wrapper does not gate layers: plain step_async semantics
            self.wrapper = reference.wrapper
            self.wrapper.init(model, parallel, 0, self.device)
            self.engine = reference.engine_cls(cache, model, parallel, get_cache_mem_alloc_backend(model.attention_backend))
        else:
            set_attention_backend(model.attention_backend)
            self.wrapper = get_attention_wrapper()
            self.wrapper.init(model, parallel, 0, self.device)
            eng = get_cache_engine(model.attention_backend)
            self.engine = eng(cache, model, parallel, get_cache_mem_alloc_backend(model.attention_backend))
        self.Hq = model.get_num_q_heads(parallel)
        self.Hkv = model.get_num_kv_heads(parallel)
        self.D = model.get_head_size()
        self.L = model.get_num_layers(parallel)
        self.scale = self.D ** -0.5
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        self._gen = g
        self._bufs = {}
        self.stats = ReplayStats()
        self.sample_kv_util = True
        self.iter_hook = None       # called once per iteration right after engine.step (bench.py: the TP control-plane exchange)
        self.next_request = None    # (seq_id, context length) — or a list of them — the scheduler will admit next: pre-mapped under this iteration's forward
        self.admission_lookahead = True
        self.time_iterations = False    # record one HIP event per iteration end (GPU time of every iteration: time-weighted KV utilisation)
        self._phase = 0
        # Compute runs on a NON-BLOCKING stream: on ROCm 7.2 hipMemMap / hipMemUnmap wait for work queued on
        # the legacy default stream (and every blocking stream) but not for non-blocking streams
        # (tools/vmm_probe.cpp, profiles/r01_vmm_sync_probe_raw.txt), so this is what lets page mapping — on the
        # mapper thread or in step_async's synchronous part — overlap the forward pass.
        self.stream = torch.cuda.Stream(device=self.device)
        from . import vattention as _va
        self._tpp = _va.layout()["tokens_per_page"]
        self.page_size = self.engine.page_size       # what the engine actually uses (its layout policy may have replaced the configured one)

    def _qkv(self, T: int):
        # synthetic N(0,1) activations, one set per token count (the transformer body is out of scope)
        b = self._bufs.get(T)
        if b is None:
            with torch.cuda.stream(self.stream):        # allocated and consumed on the compute stream
                mk = lambda h: torch.randn(T, h * self.D, generator=self._gen, device=self.device, dtype=torch.float32).to(self.model.dtype)
                b = self._bufs[T] = (mk(self.Hq), mk(self.Hkv), mk(self.Hkv))
                if len(self._bufs) > 8:
                    self._bufs.pop(next(iter(self._bufs)))
        return b

    def run_iteration(self, mds: List[SequenceMetadata]) -> torch.Tensor:
        """One scheduler iteration: prefills must precede decodes in `mds` (seq_manager.on_schedule order)."""
        T = 0
        for md in mds:
            T += md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1
        q, k, v = self._qkv(T)
        with torch.cuda.stream(self.stream):
            self.engine.step(mds)
            if self.iter_hook is not None:
                self.iter_hook(self)
            if self.next_request is not None and self.admission_lookahead and hasattr(self.engine, "prefetch_request"):
                nxt = self.next_request if isinstance(self.next_request, list) else [self.next_request]
                for sid, n in nxt:
                    self.engine.prefetch_request(sid, n)               # queued behind this step's own look-ahead batch
            self.next_request = None
            self.wrapper.begin_forward(mds)
            out = None
            for layer in range(self.L):
                out = self.wrapper.forward(q, k, v, self.engine.gpu_cache[layer], self.scale, layer)
            self.wrapper.end_forward()
        for md in mds:                               # seq_manager.on_step_completed
            if md.is_prompt:
                n = md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                c = md.seq.prompt_processed
                self.stats.prefill_pairs += n * c + n * (n + 1) / 2.0
                md.seq.prompt_processed += n
                self.stats.prefill_tokens += n
                if md.seq.prompt_done:
                    md.seq.output_len += 1           # the prefill iteration emits the first output token
                    self.stats.decode_tokens += 1
            else:
                self.stats.decode_pairs += md.seq.get_len()
                md.seq.output_len += 1
                self.stats.decode_tokens += 1
        if self.sample_kv_util:
            self._sample_util()
        elif self.time_iterations:
            self.stats.iter_util.append((None, None, 0, None))      # keep iter_util aligned with iter_events / iter_phase
        if self.time_iterations:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
            self.stats.iter_events.append(ev)
            self.stats.iter_phase.append(self._phase)
        with torch.cuda.stream(self.stream):         # free_batch_idx records the slot's fence on the stream the kernels ran on
            self.engine.on_step_completion(mds)
        self.stats.iterations += 1
        return out

    def _sample_util(self):
        from . import vattention
        c = vattention.counts()                      # O(max_batch_size), no pool dump
        tpp = self._tpp
        mapped_tokens = c["mapped_groups"] * tpp
        live = sum(self.engine.curr_seq_lens)
        u_mapped = u_needed = None
        if mapped_tokens:
            u_mapped = live / mapped_tokens
            self.stats.kv_util_samples.append(u_mapped)
        if c["needed_groups"]:
            # internal fragmentation only: pages a LIVE sequence needs vs the tokens it holds (pages kept mapped under
            # finished slots by deferred reclamation are cache, reclaimable on demand, not fragmentation)
            u_needed = live / (c["needed_groups"] * tpp)
            self.stats.kv_needed_samples.append((u_needed, c["active_slots"]))
        pages_per_group = 2 if self.engine.vattn_mega_cache else 2 * self.L
        total_pages = c["pool_pages"] + c["mapped_groups"] * pages_per_group
        if total_pages:
            self.stats.ext_frag_max = max(self.stats.ext_frag_max, (c["pool_pages"] % pages_per_group) / total_pages)
            self.stats.ext_frag_samples += 1
        reserved_tokens = (c["pool_pages"] // pages_per_group) * tpp + mapped_tokens
        if reserved_tokens:
            self.stats.mapped_over_reserved.append(mapped_tokens / reserved_tokens)
        if self.time_iterations:
            self.stats.iter_util.append((u_mapped, u_needed, c["active_slots"], mapped_tokens / reserved_tokens if reserved_tokens else None))

    def run_static_trace(self, num_requests: int, total_len: int, pd_ratio: float, chunk_size: Optional[int] = None) -> ReplayStats:
        """scripts/benchmark_e2e_static_trace.py: decode = ceil(total/(1+P:D)), prefill = total - decode
        (uniform_request_length_generator.py:12-27).  vLLM scheduler when chunk_size is None (whole prompts),
        Sarathi-style chunked prefill otherwise (run_figure_6.sh:32-33)."""
        decode = math.ceil(total_len / (1 + pd_ratio))
        prefill = total_len - decode
        waiting = [Sequence(i, prefill, total_len) for i in range(num_requests)]
        running: List[Sequence] = []
        B = self.cache_cfg.max_batch_size
        chunk = chunk_size or prefill
        while waiting or running:
            prefilling = [s for s in running if not s.prompt_done]
            if not prefilling and waiting and len(running) < B:
                s = waiting.pop(0)
                running.append(s)
                prefilling = [s]
            if waiting and len(running) < B:
                self.next_request = (waiting[0].seq_id, min(chunk, waiting[0].prompt_len))
            if prefilling:
                mds = [SequenceMetadata(prefilling[0], chunk, True)]
                if chunk_size:                       # Sarathi: piggy-back the running decodes on the prefill chunk
                    mds += [SequenceMetadata(s, 0, False) for s in running if s.prompt_done]
            else:
                mds = [SequenceMetadata(s, 0, False) for s in running]
            self.run_iteration(mds)
            running = [s for s in running if not s.is_finished()]
        return self.stats

    def run_dynamic_trace(self, num_requests: int, seed: int = 42, max_tokens: int = 32768, watermark: float = 0.01,
                          lengths: Optional[List[List[int]]] = None, qps: Optional[float] = None,
                          body_time=(0.0, 0.0)) -> dict:
        """Capacity / fragmentation stress in the shape of the reference's dynamic trace
        (scripts/benchmark_e2e_dynamic_trace.py:7-60: 256 requests, arxiv-summarisation lengths, vLLM scheduler,
        max_batch_size 256).  `lengths` = the request lengths the reference's recipe yields (fixture generated by
        oracle/gen_golden_c3_lengths.py); without it lengths are drawn from shifted log-normals fitted to the trace's marginals (prefill min 4097 / p50 7958 / p75 13186 / max 31805, decode min 105 / p50 332 /
        p75 482; scripts/artifact_asplos25/traces/arxiv_sample.csv), total capped at max_tokens.
        Arrivals: closed loop by default (every request waiting at t = 0: without the transformer body an open loop leaves the GPU
        idle).  `qps` switches to the reference's OPEN loop (sarathi/benchmark/request_generator/
        poisson_request_interval_generator.py:9-21: Python `random` seeded with `seed`, interval = min(-ln(1 - U)/qps, 3/qps)) on a
        VIRTUAL clock: an iteration advances it by the GPU time of its attention + KV work (HIP events, one synchronisation per
        iteration) plus max(body_time[0] x tokens of the iteration, body_time[1]) seconds, the stand-in for the transformer body this
        harness does not run (GEMM time per token, and the weight-streaming floor of an iteration); a request is schedulable once
        its arrival time has passed; an idle engine jumps to the next arrival.
        Admission is the reference's count-based rule (vattention_block_space_manager.py:36-66):
        free - promised - needed >= watermark."""
        import random
        from . import vattention
        rng = random.Random(seed)
        lay = vattention.layout()
        tpp = lay["tokens_per_page"]
        pages = lambda n: (n + tpp - 1) // tpp
        reqs = []
        for i in range(num_requests):
            if lengths:        # (prefill, decode) pairs produced by the reference's own recipe: tests/golden/c3_arxiv_lengths_256.json
                pre, dec = lengths[i % len(lengths)]
            else:
                pre = min(31805, int(4097 + rng.lognormvariate(math.log(3861), 1.27)))
                dec = int(105 + rng.lognormvariate(math.log(227), 0.75))
            tot = min(max_tokens, pre + dec)
            pre = min(pre, tot - 1)
            reqs.append(Sequence(i, pre, tot))
        arrival = [0.0] * num_requests
        if qps:
            arng = random.Random(seed)        # the generator's own stream (poisson_request_interval_generator.py:12)
            t_arr = 0.0
            for i in range(num_requests):      # synthetic_request_generator.py: arrived_at = last_arrived_at + interval, from t = 0
                t_arr += min(-math.log(1.0 - arng.random()) / qps, 3.0 / qps)
                arrival[i] = t_arr
        waiting, running = list(reqs), []
        B = self.cache_cfg.max_batch_size
        total_groups = None
        out = {"peak_running": 0, "util_at_peak": [], "iters": 0, "preempted": 0}
        vm0 = vattention.stats()
        import time as _t
        t0 = _t.perf_counter()
        self.stats.iter_events, self.stats.iter_phase, self.stats.iter_util = [], [], []
        timed = self.time_iterations or bool(qps)
        keep_ti = self.time_iterations
        self.time_iterations = timed
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(self.stream)
        clock, last_ev = 0.0, ev0           # open loop: virtual seconds
        finish_at, sync_ms_per_iter, prev_sync_ns = {}, [], vm0["sync_ns"]
        while waiting or running:
            if qps and not running and waiting and arrival[waiting[0].seq_id] > clock:
                clock = arrival[waiting[0].seq_id]                       # idle engine: jump to the next arrival
            free = vattention.num_free_kvblocks()
            if free >= (1 << 63):
                free -= 1 << 64
            if total_groups is None:
                total_groups = free
            wm = int(watermark * total_groups)
            promised = sum(pages(s.total_len) - pages(max(s.get_len(), 1)) for s in running)
            # vLLM scheduler (sarathi-lean/sarathi/core/scheduler/vllm_scheduler.py): prefills are prioritised; whole prompts
            # are batched into one iteration while they fit max_tokens_in_batch (= max_tokens here) and pass admission
            admitted, budget = [], max_tokens
            while waiting and len(running) < B:
                s = waiting[0]
                if arrival[s.seq_id] > clock:
                    break
                if s.prompt_len > budget or free - promised - pages(s.total_len) < wm:
                    break
                waiting.pop(0)
                running.append(s)
                admitted.append(s)
                budget -= s.prompt_len
                promised += pages(s.total_len)
            if admitted:
                mds = [SequenceMetadata(s, s.prompt_len, True) for s in admitted]
            elif running:
                mds = [SequenceMetadata(s, 0, False) for s in running]
            else:
                raise RuntimeError("request %d (%d tokens) cannot be admitted into an EMPTY engine: the pool is too small for it" % (
                    waiting[0].seq_id, waiting[0].total_len))
            # what the NEXT iteration will admit (same token budget and batch rule, free-block rule left to the real admission): the
            # mapper thread maps these prompts' pages while this iteration computes
            nxt, bud = [], max_tokens
            for s in waiting:
                if len(running) + len(nxt) >= B or s.prompt_len > bud or arrival[s.seq_id] > clock:
                    break
                nxt.append((s.seq_id, s.prompt_len))
                bud -= s.prompt_len
            self.next_request = nxt or None
            self._phase = 0 if waiting else 1
            tokens_it = sum(md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1 for md in mds)
            self.run_iteration(mds)
            out["iters"] += 1
            if qps:
                ev = self.stats.iter_events[-1]
                ev.synchronize()
                clock += last_ev.elapsed_time(ev) * 1e-3 + max(body_time[0] * tokens_it, body_time[1])
                last_ev = ev
                ns = vattention.stats()["sync_ns"]
                sync_ms_per_iter.append((ns - prev_sync_ns) / 1e6)
                prev_sync_ns = ns
                for s in running:
                    if s.is_finished():
                        finish_at[s.seq_id] = clock
            running = [s for s in running if not s.is_finished()]
            if len(running) > out["peak_running"]:
                out["peak_running"] = len(running)
            if self.stats.kv_util_samples and len(running) >= min(B, num_requests) * 3 // 4:
                out["util_at_peak"].append(self.stats.kv_util_samples[-1])
        torch.cuda.synchronize()
        self.time_iterations = keep_ti
        out["seconds"] = _t.perf_counter() - t0
        vm1 = vattention.stats()
        u = self.stats.kv_util_samples
        out.update({
            "requests": num_requests, "tokens": self.stats.prefill_tokens + self.stats.decode_tokens,
            "tokens_per_s": (self.stats.prefill_tokens + self.stats.decode_tokens) / out["seconds"],
            "kv_live_over_mapped_mean": sum(u) / max(1, len(u)),
            "kv_live_over_mapped_at_high_concurrency": (sum(out["util_at_peak"]) / len(out["util_at_peak"])) if out["util_at_peak"] else None,
            "mapped_over_pool_max": max(self.stats.mapped_over_reserved) if self.stats.mapped_over_reserved else None,
            "kv_live_over_needed_mean": (sum(x for x, _ in self.stats.kv_needed_samples) / max(1, len(self.stats.kv_needed_samples))),
            "kv_live_over_needed_at_peak": min((x for x, a in self.stats.kv_needed_samples if a >= 0.9 * out["peak_running"]), default=None),
            "slack_tokens_per_seq_at_peak_max": max(((1 - x) * 1.0 for x, a in self.stats.kv_needed_samples if a >= 0.9 * out["peak_running"]), default=None),
            "tokens_per_page": tpp,
            "external_fragmentation_max": self.stats.ext_frag_max, "external_fragmentation_samples": self.stats.ext_frag_samples,
            "map_calls": vm1["map_calls"] - vm0["map_calls"], "unmap_calls": vm1["unmap_calls"] - vm0["unmap_calls"],
            "sync_map_ms": (vm1["sync_ns"] - vm0["sync_ns"]) / 1e6, "async_map_ms": (vm1["async_ns"] - vm0["async_ns"]) / 1e6,
            "join_wait_ms": (vm1["join_wait_ns"] - vm0["join_wait_ns"]) / 1e6,
            "tlb_flushes": vm1["tlb_flushes"] - vm0["tlb_flushes"], "tlb_flush_ms": (vm1["tlb_flush_ns"] - vm0["tlb_flush_ns"]) / 1e6,
            "handles_created": vm1["handles_created"] - vm0["handles_created"], "create_ms": round((vm1["create_ns"] - vm0["create_ns"]) / 1e6, 1),
            "fence_waits": vm1["fence_waits"] - vm0["fence_waits"], "quiesce_calls": vm1["quiesce_calls"] - vm0["quiesce_calls"],
            "sync_breakdown": {"batches": vm1["sync_batches"] - vm0["sync_batches"], "maps": vm1["sync_maps"] - vm0["sync_maps"],
                               "unmaps": vm1["sync_unmaps"] - vm0["sync_unmaps"], "creates": vm1["sync_creates"] - vm0["sync_creates"],
                               "create_ms": round((vm1["sync_create_ns"] - vm0["sync_create_ns"]) / 1e6, 1),
                               "fence_ms": round((vm1["sync_fence_ns"] - vm0["sync_fence_ns"]) / 1e6, 1),
                               "tlb_ms": round((vm1["sync_tlb_ns"] - vm0["sync_tlb_ns"]) / 1e6, 1)},
        })
        out.pop("util_at_peak")
        if timed and self.stats.iter_events:
            out["kv_util_time_weighted"] = self._time_weighted_util(ev0)
        if qps:
            lat = sorted((finish_at[r.seq_id] - arrival[r.seq_id]) / r.total_len for r in reqs if r.seq_id in finish_at)
            sm = sorted(sync_ms_per_iter)
            pct = lambda xs, q: xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None
            out["open_loop"] = {"qps": qps, "virtual_seconds": round(clock, 3), "body_seconds_per_token": body_time[0], "body_seconds_floor_per_iteration": body_time[1],
                                "request_e2e_time_normalized_p50": pct(lat, 0.5), "request_e2e_time_normalized_p99": pct(lat, 0.99),
                                "sync_map_ms_per_step_p50": pct(sm, 0.5), "sync_map_ms_per_step_p99": pct(sm, 0.99),
                                "sync_map_ms_per_step_max": sm[-1] if sm else None}
        return out

    def _time_weighted_util(self, ev0) -> dict:
        """KV utilisation weighted by the GPU time of every iteration (HIP events), separately for the window in which requests are
        still waiting to be admitted (steady state: the pool is the contended resource) and for the drain tail (no admissions left:
        finished slots keep their pages under deferred reclamation because nobody needs them)."""
        evs, ph, ut = self.stats.iter_events, self.stats.iter_phase, self.stats.iter_util
        assert len(evs) == len(ph) == len(ut), "per-iteration event / phase / utilisation lists out of step"
        acc = {0: [0.0, 0.0, 0.0, 0.0, 0.0], 1: [0.0, 0.0, 0.0, 0.0, 0.0]}      # time, t*live/mapped, t*live/needed, t*mapped/pool, t with samples
        prev = ev0
        for ev, phase, u in zip(evs, ph, ut):
            dt = prev.elapsed_time(ev)
            prev = ev
            a = acc[phase]
            a[0] += dt
            if u[0] is not None and u[1] is not None:
                a[1] += dt * u[0]
                a[2] += dt * u[1]
                a[3] += dt * (u[3] or 0.0)
                a[4] += dt
        def summ(a):
            return None if a[4] == 0 else {"gpu_ms": round(a[0], 1), "live_over_mapped": round(a[1] / a[4], 4),
                                           "live_over_needed": round(a[2] / a[4], 4), "mapped_over_pool": round(a[3] / a[4], 4)}
        both = [x + y for x, y in zip(acc[0], acc[1])]
        return {"steady_state": summ(acc[0]), "drain_tail": summ(acc[1]), "whole_run": summ(both)}

    def close(self):
        self.engine.cleanup_kvcache()
