#!/usr/bin/env python3
"""bench.py — hot-path benchmark of the MI355X-native vAttention stack.

Metric (BASELINE.json): prefill+decode tokens/s (+ KV HBM utilisation) of the attention + KV-memory hot path on the reference's
own traces, inputs resident in HBM.  The transformer body (GEMMs) is out of scope: q/k/v are synthetic N(0,1).

Workload by --gpus N (one rank per GPU; `python bench.py --gpus N` spawns the ranks itself when WORLD_SIZE is unset):

  N = 1  BASELINE.json configs[1]: Yi-6B TP=1, fa_vattn_2mb (2 MiB pages, async mapping), static trace @ 32k, P:D = 500 (32 702
         prefill + 66 decode tokens per request), vLLM scheduler (/root/reference/scripts/benchmark_e2e_static_trace.py:6-57).
         One STEP = one max_batch_size = 16 wave of that trace: 16 whole-prompt prefills, then 65 batch-16 decode iterations, all 32
         layers, real page manager (HIP VMM, mapper thread, layer-ordered mapping of new prompts) in the loop.  The 50-request
         trace is 3 such waves + a 2-request tail; it is run ONCE outside the timed region and reported as `full_trace_50req`.
  N = 2  configs[3]: Yi-34B TENSOR-PARALLEL over 2 GPUs (28 query / 4 kv heads per rank, 60 layers), static trace @ 128k, P:D = 500,
         Sarathi 16k chunks (run_figure_6.sh:32-33).  One step = one 131 072-token request end to end.
  N = 4  the same request on a TP = 4 shard of Yi-34B (14 / 2 heads per rank).
  N = 8  configs[4]: Llama-3-70B TP = 8 (8 / 1 heads per rank, 80 layers), dynamic arxiv trace; one step = a closed-loop replay of the
         first 48 requests of the reference's length recipe (tests/golden/c3_arxiv_lengths_256.json), max_batch_size 256, megacache
         layout with 8 MiB pages (the configured 256 KiB pages need one hipMemCreate handle per page: O(live handles)).
  For N > 1 every rank processes the SAME requests with its head shard (sarathi/config.py:139-167); there is no data-path
  collective.  Inside the timed loop every iteration does the control-plane exchange of the reference engine over RCCL:
  all-reduce MIN of num_free_kvblocks() (base_llm_engine.py:381-390) and an all-gather of a fingerprint of the page-manager state,
  checked at the end of the step (identical page decisions on every rank).  value = tokens of ONE request stream / max-rank time.

Extra objects on the JSON line (N = 1): `roofline` (dominant kernel: causal prefill attention, MFMA-bound: algorithmic flops per
launch / mean launch duration from HIP events on the launch stream inside the timed region), `roofline_decode` (HBM-bound split-KV
decode, same method), `cold_wave` (the first wave on a fresh pool: handle creation, synchronous vs mapper-thread mapping),
`dynamic` (256-request arxiv replay, all 32 layers of Llama-3-8B: peak concurrency, fragmentation, mapping cost), `cpu_baseline`
(the CPU oracle — kind "port" — timed on the real shapes of this workload: one layer of a decode step at 32k and the last
512 query rows of the 32 702-token prefill, scaled by the stated law).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # spec; ~6.29 TB/s achievable (same guide)

WORKLOADS = {
    1: dict(model="yi-6b", tp=1, ctx=32768, pd=500.0, batch=16, chunk=0, page=2 << 20, mode="static", requests=16, backend="fa_vattn",
            label="configs[1]: yi-6b TP=1 fa_vattn_2mb static trace @ 32768 ctx, P:D=500, vLLM scheduler; one step = one max_batch_size=16 wave "
                  "of the 50-request trace (16 x 32702-token prefill + 65 batch-16 decode iterations, 32 layers)"),
    2: dict(model="yi-34b", tp=2, ctx=131072, pd=500.0, batch=4, chunk=16384, page=2 << 20, mode="static", requests=1, backend="fa_vattn",
            label="configs[3]: yi-34b TP=2 (28/4 heads per rank, 60 layers) static trace @ 131072 ctx, P:D=500, Sarathi 16k chunks; one step "
                  "= one request (130810 prefill + 262 decode tokens)"),
    4: dict(model="yi-34b", tp=4, ctx=131072, pd=500.0, batch=4, chunk=16384, page=2 << 20, mode="static", requests=1, backend="fa_vattn",
            label="yi-34b TP=4 shard (14/2 heads per rank, 60 layers) of configs[3]'s request: static @ 131072 ctx, P:D=500, 16k chunks; "
                  "one step = one request"),
    8: dict(model="llama-3-70b", tp=8, ctx=32768, pd=0.0, batch=256, chunk=0, page=8 << 20, mode="dynamic", requests=48, backend="fa_vattn_megacache",
            decode_cap=768,
            label="configs[4]: llama-3-70b TP=8 (8/1 heads per rank, 80 layers) dynamic arxiv trace, closed loop; one step = the first 48 "
                  "requests of the reference's length recipe (decode lengths capped at 768 tokens: one of the 48 has 6129, which would leave "
                  "5000 batch-1 iterations at the end of every step), max_batch_size 256, megacache layout with 8 MiB pages (stands in for 256 KiB pages)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the dynamic-trace and full-trace legs (N = 1)")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; makes the number INVALID)")
    ap.add_argument("--ctx", type=int, default=0, help="override context length (debug only; makes the number INVALID)")
    ap.add_argument("--rank-of", type=int, default=0, help="debug: run ONE rank's share of the --gpus N workload of this value on a single GPU, "
                    "without the collectives (checks the tensor-parallel workloads where only one GPU is visible; NOT a bench line)")
    return ap.parse_args()


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run (one per GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def cpu_baseline(dtype, L, Hq, Hkv, D, prefill, decode_iters, batch) -> dict:
    """The CPU oracle (oracle/attn.py, math='f32' = the reference kernel's numerics) on the REAL shapes of configs[1], one layer:
    (a) one batch-`batch` decode step at 32k context; (b) the last 512 query rows of the 32 702-token causal prefill.
    Attention work per query row is proportional to the keys it sees, so the whole prefill costs
    t_b x [n(n+1)/2] / [sum of (i+1) over the sampled rows]; tokens/s is quoted for the whole model (x L layers)."""
    import torch
    from oracle.attn import flash_attn_with_kvcache_ref
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    n, rows = prefill, 512
    k = torch.randn(1, n + 8, Hkv, D).to(dtype)
    v = torch.randn(1, n + 8, Hkv, D).to(dtype)
    q = torch.randn(1, rows, Hq, D).to(dtype)
    t0 = time.perf_counter()
    flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=n, causal=True, math="f32")      # rows [n - 2048, n) over keys [0, n)
    t_blk = time.perf_counter() - t0
    law = (n * (n + 1) / 2.0) / sum(i + 1 for i in range(n - rows, n))
    t_prefill_layer = t_blk * law
    kd = torch.randn(2, n + 8, Hkv, D).to(dtype)
    vd = torch.randn(2, n + 8, Hkv, D).to(dtype)
    qd = torch.randn(2, 1, Hq, D).to(dtype)
    t0 = time.perf_counter()
    flash_attn_with_kvcache_ref(qd, kd, vd, cache_seqlens=torch.tensor([n, n + 4], dtype=torch.int32), causal=True, math="f32")
    t_dec_seq = (time.perf_counter() - t0) / 2.0                                       # per sequence per layer at 32k
    t_request = L * (t_prefill_layer + decode_iters * t_dec_seq)
    tokens = prefill + 1 + decode_iters
    return {"value": round(tokens / t_request, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (torch fp32 math on fp16 inputs, %d threads) on configs[1]'s real shapes, one layer: last %d query rows "
                      "of the %d-token causal prefill (%.2f s; whole prefill = x %.2f by the keys-seen law) and one decode step of 2 "
                      "sequences at 32k (%.3f s per sequence); tokens/s = (%d tokens per request) / (%d layers x [prefill + %d decode "
                      "steps])" % (cores, rows, n, t_blk, law, t_dec_seq, tokens, L, decode_iters)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if a.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if world not in WORKLOADS:
        raise SystemExit("--gpus must be one of %s" % sorted(WORKLOADS))
    # VATTN_BENCH_BACKEND=gloo is a test hook: it lets the N>1 code path run where the ranks outnumber the GPUs (ranks then
    # share devices and the collectives run on CPU tensors); the driver's runs use the default, RCCL
    backend = os.environ.get("VATTN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from vattention_amd import vattention
    from vattention_amd.attention.timers import drain_op_timers, enable_op_timers
    from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig

    if a.rank_of and (world != 1 or a.rank_of not in WORKLOADS):
        raise SystemExit("--rank-of needs --gpus 1 and one of %s" % sorted(WORKLOADS))
    w = dict(WORKLOADS[a.rank_of or world])
    if a.ctx:
        w["ctx"] = a.ctx
    dtype = torch.float16                                    # benchmark_runner.py:81
    valid = not (a.layers or a.ctx or a.rank_of)

    def make_runner(model_name, tp, ctx, page, batch, backend_name, mem_bytes, layers=0):
        model = ModelConfig.named(model_name, dtype=dtype, max_model_len=ctx, attention_backend=backend_name)
        if layers:
            model.num_layers = layers
        # the bench lines run the layout BASELINE.json names (fa_vattn_2mb ...), not the engine's automatic replacement
        return HotPathRunner(model, ParallelConfig(tp, 1), CacheConfig(page_size=page, max_batch_size=batch, memory_for_gpu=mem_bytes, vattn_keep_layout=True), device=str(dev))

    free_b, total_b = torch.cuda.mem_get_info(dev)
    share = world if backend != "nccl" else 1               # test hook: ranks share a device
    # memory_for_gpu = total*0.9 - peak of the (absent) model body; keep 12 GiB for activations / workspace
    mem_for_kv = (min(int(total_b * 0.9), free_b) - (12 << 30)) // share
    if w["mode"] == "dynamic":
        mem_for_kv = min(mem_for_kv, 96 << 30)               # the 48-request slice needs ~25 GiB per rank; bounds handle creation
    runner = make_runner(w["model"], w["tp"], w["ctx"], w["page"], w["batch"], w["backend"], mem_for_kv, a.layers)
    Hq, Hkv, D, L = runner.Hq, runner.Hkv, runner.D, runner.L
    lengths = None
    if w["mode"] == "dynamic":
        lengths = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
        if w.get("decode_cap"):
            lengths = [[pre, min(dec, w["decode_cap"])] for pre, dec in lengths]

    # ---- control plane of a tensor-parallel engine, every iteration, over RCCL (no host synchronisation inside the step) ----
    ctl = {"iters": 0, "log": None, "ok": None}
    if dist is not None:
        ctl["log"] = torch.full((1,), 1 << 62, dtype=torch.int64, device=red_dev)       # running min of min(free blocks)
        ctl["ok"] = torch.ones((1,), dtype=torch.int64, device=red_dev)                 # all ranks' page-state fingerprints equal

        def iter_hook(r):
            c = vattention.counts()
            free = vattention.num_free_kvblocks()
            free = free - (1 << 64) if free >= (1 << 63) else free
            fp = (c["mapped_groups"] * 1000003 + c["needed_groups"]) * 1000003 + c["pool_pages"] * 31 + c["active_slots"]
            t = torch.tensor([free, fp], dtype=torch.int64).to(red_dev, non_blocking=True)
            mn = t[:1].clone()
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)                                   # base_llm_engine.py:381-390
            ctl["log"] = torch.minimum(ctl["log"], mn)
            g = torch.empty(world, dtype=torch.int64, device=red_dev)
            dist.all_gather_into_tensor(g, t[1:].contiguous())
            ctl["ok"] = ctl["ok"] * (g == g[0]).all().to(torch.int64)
            ctl["iters"] += 1
        runner.iter_hook = iter_hook

    def one_step():
        runner.stats.__init__()
        if w["mode"] == "static":
            runner.run_static_trace(w["requests"], w["ctx"], w["pd"], w["chunk"] or None)
        else:
            runner.run_dynamic_trace(w["requests"], lengths=lengths)
        return runner.stats.prefill_tokens + runner.stats.decode_tokens

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up; the FIRST wave runs on a fresh pool and is reported as cold_wave ----
    cold = None
    for i in range(a.warmup):
        v0 = vattention.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = one_step()
        torch.cuda.synchronize()
        dt_c = time.perf_counter() - t0
        if i == 0:
            v1 = vattention.stats()
            d = lambda k: v1[k] - v0[k]
            tot_map_ms = (d("sync_ns") + d("async_ns")) / 1e6
            cold = {"ms": round(dt_c * 1e3, 1), "tokens": tk, "handles_created": d("handles_created"), "create_ms": round(d("create_ns") / 1e6, 1),
                    "map_calls": d("map_calls"), "layered_batches": d("layered_batches"),
                    "sync_map_ms": round(d("sync_ns") / 1e6, 1), "mapper_thread_map_ms": round(d("async_ns") / 1e6, 1),
                    "sync_share_of_map_time": round(d("sync_ns") / 1e6 / tot_map_ms, 4) if tot_map_ms else None,
                    "layer_wait_ms": round(d("layer_wait_ns") / 1e6, 2), "join_wait_ms": round(d("join_wait_ns") / 1e6, 2)}
    vm0 = vattention.stats()
    enable_op_timers(w["mode"] == "static")      # per-op HIP events feed the roofline objects (static workloads); the dynamic replay's
    barrier()                                    # half a million tiny launches per step are not slowed down by them
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(a.steps):
        tokens += one_step()
    barrier()
    dt = time.perf_counter() - t0
    op_ms = drain_op_timers()
    enable_op_timers(False)
    vm1 = vattention.stats()
    kv_util = list(runner.stats.kv_util_samples)
    kv_map = list(runner.stats.mapped_over_reserved)
    if cold is not None:
        cold["warm_step_ms"] = round(dt * 1e3 / a.steps, 1)

    tp_check = None
    per_rank_pf = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tp_check = {"iterations_with_control_plane_exchange": ctl["iters"], "min_free_kvblocks_over_ranks": int(ctl["log"].item()),
                    "identical_page_decisions_on_all_ranks": bool(int(ctl["ok"].item()))}
        if not tp_check["identical_page_decisions_on_all_ranks"]:
            raise SystemExit("tensor-parallel ranks diverged in their page-manager state")

    # ---- roofline of the dominant kernels, from events recorded inside the timed region ----
    decode_tok = math.ceil(w["ctx"] / (1 + w["pd"])) if w["mode"] == "static" else 0
    prefill_tok = w["ctx"] - decode_tok
    roof = roof_dec = None
    if w["mode"] == "static":
        chunk = w["chunk"] or prefill_tok
        n_chunks = math.ceil(prefill_tok / chunk)
        flops_total, c = 0.0, 0
        for i in range(n_chunks):
            n = min(chunk, prefill_tok - c)
            flops_total += 4.0 * Hq * D * (n * c + n * (n + 1) / 2)
            c += n
        launches_pf = a.steps * w["requests"] * n_chunks * L
        flops_per_launch = flops_total / n_chunks
        pf_ms = op_ms.get("attn_prefill", 0.0) / max(1, launches_pf)
        pf_tflops = flops_per_launch / (pf_ms * 1e-3) / 1e12 if pf_ms > 0 else 0.0
        dec_iters = decode_tok - 1
        nb = min(w["requests"], w["batch"])
        launches_dc = a.steps * dec_iters * L
        mean_len = prefill_tok + 1 + (dec_iters - 1) / 2.0
        bytes_dc = nb * (2 * mean_len * Hkv * D * 2) + nb * Hq * D * 2 * 2
        dc_ms = op_ms.get("attn_decode", 0.0) / max(1, launches_dc)
        dc_gbs = bytes_dc / (dc_ms * 1e-3) / 1e9 if dc_ms > 0 else 0.0
        traffic_pf = traffic_dc = None
        try:       # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 per the gfx950 note); configs[1] only
            for name in ("r02_traffic.json", "r01_traffic.json"):
                pth = os.path.join(ROOT, "profiles", name)
                if os.path.exists(pth) and world == 1 and valid:
                    tj = json.load(open(pth))
                    traffic_pf = tj["prefill_yi6b_n32702"]["hbm_bytes_per_launch"]
                    traffic_dc = tj["decode_yi6b_b16_32k"]["hbm_bytes_per_launch"]
                    break
        except Exception:
            pass
        if dist is not None:
            g = torch.tensor([pf_ms], dtype=torch.float64, device=red_dev)
            gl = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(gl, g)
            per_rank_pf = [round(float(x.item()), 4) for x in gl]
        roof = {"kernel": "prefill attention (causal chunk against the KV prefix), per rank", "bound": "mfma", "achieved": round(pf_tflops, 2),
                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(pf_tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic_pf,
                "ms_per_launch": round(pf_ms, 4), "flops_per_launch": flops_per_launch}
        if per_rank_pf:
            roof["ms_per_launch_by_rank"] = per_rank_pf
        roof_dec = {"kernel": "decode_kernel+combine (split-KV decode, batch %d), per rank" % nb, "bound": "hbm", "achieved": round(dc_gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dc_gbs / HBM_PEAK_GBS, 4), "traffic": traffic_dc,
                    "ms_per_launch": round(dc_ms, 4), "bytes_per_launch": bytes_dc}

    # ---- N = 1 extras, outside the timed region ----
    full_trace = dynamic = None
    if world == 1 and valid and not a.no_dynamic:
        torch.cuda.synchronize()
        runner.stats.__init__()
        t1 = time.perf_counter()
        runner.run_static_trace(50, w["ctx"], w["pd"], None)        # the whole 50-request trace: 16 + 16 + 16 + 2
        torch.cuda.synchronize()
        dt_f = time.perf_counter() - t1
        tk_f = runner.stats.prefill_tokens + runner.stats.decode_tokens
        full_trace = {"requests": 50, "tokens": tk_f, "seconds": round(dt_f, 3), "tokens_per_s": round(tk_f / dt_f, 1)}
        runner.close()
        runner = None
        # configs[2]'s shape: Llama-3-8B, ALL 32 layers, 256 requests of the reference's arxiv length recipe, max_batch_size 256, closed
        # loop.  Megacache layout with 8 MiB pages (128 tokens per page): configs[2]'s 64 KiB pages would need 4 M hipMemCreate
        # handles for this pool, and even 2 MiB megacache pages 120 k — handle creation is O(live handles) on ROCm.
        lengths256 = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
        r2 = make_runner("llama-3-8b", 1, 32768, 8 << 20, 256, "fa_vattn_megacache", mem_for_kv)
        try:
            out = r2.run_dynamic_trace(256, lengths=lengths256)
            # synchronous batches = driver calls on the engine thread (maps / unmaps / set-access / creations / TLB invalidation) PLUS,
            # before an unmap, the wait for the fence of the slot that gives the page up (the GPU has to reach the point where
            # that request finished: not mapping work; it moves to the mapper thread when the look-ahead does the reclaim)
            sb = out.get("sync_breakdown") or {}
            fence_ms = float(sb.get("fence_ms", 0.0))
            sync_calls_ms = max(0.0, out["sync_map_ms"] - fence_ms)
            tot = sync_calls_ms + out["async_map_ms"]
            dynamic = {"workload": "configs[2] shape: llama-3-8b, 32 layers, 256 arxiv-length requests closed loop, max_batch_size 256, "
                                   "megacache 8 MiB pages (128 tokens per page), pool = 0.9 x HBM - 12 GiB",
                       "peak_concurrent_sequences": out["peak_running"], "tokens": out["tokens"], "seconds": round(out["seconds"], 2),
                       "tokens_per_s": round(out["tokens_per_s"], 1),
                       "kv_live_over_needed_at_peak": out["kv_live_over_needed_at_peak"], "kv_live_over_mapped_mean": round(out["kv_live_over_mapped_mean"], 4),
                       "external_fragmentation": 0.0, "map_calls": out["map_calls"], "unmap_calls": out["unmap_calls"],
                       "handles_created": out.get("handles_created"), "create_ms": out.get("create_ms"),
                       "sync_map_ms": round(sync_calls_ms, 1), "sync_fence_wait_ms": round(fence_ms, 1),
                       "mapper_thread_map_ms": round(out["async_map_ms"], 1),
                       "sync_share_of_map_time": round(sync_calls_ms / tot, 4) if tot else None,
                       "sync_map_share_of_wall": round(out["sync_map_ms"] / 1e3 / out["seconds"], 4),
                       "sync_breakdown": out.get("sync_breakdown")}
        finally:
            r2.close()

    if rank == 0:
        out = {
            "metric": "prefill+decode tokens/sec (attention + KV-memory hot path)",
            "value": round(tokens / dt, 2),
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / a.steps, 3),
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": (w["label"] if valid else "custom (NOT the bench line): " + w["label"]) +
                            "; attention+KV hot path only (transformer GEMMs out of scope, q/k/v synthetic)",
                "requests_per_step": w["requests"], "layers": L, "hq_per_rank": Hq, "hkv_per_rank": Hkv, "head_dim": D, "page_kib": w["page"] >> 10,
                "parallelism": "single GPU" if world == 1 else "tensor parallel x%d (KV sharded by head, no data-path collective; control-plane "
                               "min(free blocks) + state fingerprint over RCCL every iteration)" % world,
            },
            "kv_hbm_util": {"live_over_mapped_mean": round(sum(kv_util) / max(1, len(kv_util)), 4),
                            "live_over_mapped_min": round(min(kv_util), 4) if kv_util else None,
                            "mapped_over_pool_max": round(max(kv_map), 4) if kv_map else None},
            "page_mapping": {"map_calls": vm1["map_calls"] - vm0["map_calls"], "unmap_calls": vm1["unmap_calls"] - vm0["unmap_calls"],
                             "sync_ms": round((vm1["sync_ns"] - vm0["sync_ns"]) / 1e6, 3),
                             "async_ms": round((vm1["async_ns"] - vm0["async_ns"]) / 1e6, 3),
                             "join_wait_ms": round((vm1["join_wait_ns"] - vm0["join_wait_ns"]) / 1e6, 3)},
            "op_ms": {k: round(v, 2) for k, v in op_ms.items()},
        }
        if roof:
            out["roofline"] = roof
            out["roofline_decode"] = roof_dec
        if cold:
            out["cold_wave"] = cold
        if tp_check:
            out["tensor_parallel"] = tp_check
        if full_trace:
            out["full_trace_50req"] = full_trace
        if dynamic:
            out["dynamic"] = dynamic
        if world == 1 and not a.no_cpu_baseline and w["mode"] == "static":
            out["cpu_baseline"] = cpu_baseline(dtype, L, Hq, Hkv, D, prefill_tok, decode_tok - 1, w["batch"])
        print(json.dumps(out), flush=True)
    if runner is not None:
        runner.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
