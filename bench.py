#!/usr/bin/env python3
"""bench.py — hot-path benchmark of the MI355X-native vAttention stack.

Metric (BASELINE.json): prefill+decode tokens/s (+ KV HBM utilisation) on the reference's own static
trace.  Workload at N=1 = BASELINE.json configs[1]: Yi-6B TP=1, `fa_vattn_2mb` backend (2 MiB pages,
async mapping), static trace @ 32k context, P:D = 500 (32 702 prefill + 66 decode tokens per
request), vLLM scheduler, max_batch_size 16 (/root/reference/scripts/benchmark_e2e_static_trace.py:6-57).

One STEP = one full scheduler wave of that trace on one GPU: 16 requests (= max_batch_size) admitted,
each prefilled whole (32 702 tokens x 32 layers: cache_flat + causal prefill attention), then decoded
together to completion (65 iterations x 32 layers of batch-16 split-KV decode with in-kernel KV
append), with the real page manager (HIP VMM map/unmap, mapper thread) in the loop, then freed.
The transformer body (GEMMs) is out of scope: q/k/v are synthetic N(0,1), so
`value` is the tokens/s of the attention + KV-memory hot path, inputs resident in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank replays the same wave on its
own requests — the path shards by request (and by KV head under TP) with no data-path collective, so
scaling is weak and `value` = total tokens of all ranks / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel = chunked/whole-prompt causal prefill attention,
MFMA-bound: algorithmic flops per launch / mean launch duration measured with HIP events on the launch
stream inside the timed region), `roofline_decode` (HBM-bound split-KV decode, same method),
`cpu_baseline` (the CPU oracle — kind "port" — on a bounded sample of the same workload, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # spec; ~6.29 TB/s achievable (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="yi-6b")
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--pd-ratio", type=float, default=500.0)
    ap.add_argument("--batch", type=int, default=16, help="max_batch_size = requests per wave")
    ap.add_argument("--page-size", type=int, default=2 << 20)
    ap.add_argument("--mem-util", type=float, default=0.9)
    ap.add_argument("--chunk", type=int, default=0, help="0 = vLLM scheduler (whole prompts); >0 = chunked prefill")
    ap.add_argument("--tp", type=int, default=1, help="run ONE rank's share of a tensor-parallel model (heads / tp); other configs than the default are not the bench line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; makes the number INVALID)")
    return ap.parse_args()


def cpu_baseline(model_name: str, dtype) -> dict:
    """The CPU oracle (oracle/attn.py, math='f32' = the reference kernel's numerics) on a bounded sample:
    ONE layer of one request: causal prefill of the first 2048 prompt tokens and 4 decode steps at 4096
    context.  tokens/s is reported per full model (sample time x num_layers)."""
    from oracle.attn import flash_attn_with_kvcache_ref
    from vattention_amd.replay import MODELS
    L, Hq, Hkv, D = MODELS[model_name]
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    n = 2048
    q = torch.randn(1, n, Hq, D).to(dtype)
    k = torch.randn(1, 4096 + 8, Hkv, D).to(dtype)
    v = torch.randn(1, 4096 + 8, Hkv, D).to(dtype)
    t0 = time.perf_counter()
    flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=n, causal=True, math="f32")
    t_pre = time.perf_counter() - t0
    qd = torch.randn(1, 1, Hq, D).to(dtype)
    t0 = time.perf_counter()
    nd = 4
    for i in range(nd):
        flash_attn_with_kvcache_ref(qd, k, v, cache_seqlens=4096 + i, causal=True, math="f32")
    t_dec = time.perf_counter() - t0
    tokens = n + nd
    tps = tokens / ((t_pre + t_dec) * L)
    return {"value": round(tps, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (torch fp32 math on fp16 inputs, %d threads): one layer x [causal prefill of %d tokens "
                      "(%.2fs) + %d decode steps at 4096 ctx (%.2fs)], scaled by %d layers; note attention cost grows "
                      "~quadratically with context, the GPU number is at 32k" % (cores, n, t_pre, nd, t_dec, L)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # VATTN_BENCH_BACKEND=gloo is a test hook: it lets the N>1 code path run where the ranks outnumber the GPUs (ranks then
    # share devices and the timing reduction runs on CPU tensors); the driver's runs use the default, RCCL
    backend = os.environ.get("VATTN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
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

    dtype = torch.float16                                    # benchmark_runner.py:81
    model = ModelConfig.named(a.model, dtype=dtype, max_model_len=a.ctx, attention_backend="fa_vattn")
    if a.layers:
        model.num_layers = a.layers
    par = ParallelConfig(a.tp, 1)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    # memory_for_gpu = total*util - peak of the (absent) model body; keep 12 GiB for activations/workspace
    mem_for_kv = min(int(total_b * a.mem_util), free_b) - (12 << 30)
    cache = CacheConfig(page_size=a.page_size, max_batch_size=a.batch, memory_for_gpu=mem_for_kv)
    runner = HotPathRunner(model, par, cache, device=str(dev))
    Hq, Hkv, D, L = runner.Hq, runner.Hkv, runner.D, runner.L
    decode = math.ceil(a.ctx / (1 + a.pd_ratio))
    prefill = a.ctx - decode
    chunk = a.chunk or None

    def one_step():
        runner.stats.__init__()
        runner.run_static_trace(a.batch, a.ctx, a.pd_ratio, chunk)
        return runner.stats.prefill_tokens + runner.stats.decode_tokens

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    vm0 = vattention.stats()
    enable_op_timers(True)
    barrier()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(a.steps):
        tokens += one_step()
    barrier()
    dt = time.perf_counter() - t0
    op_ms = drain_op_timers()
    enable_op_timers(False)
    vm1 = vattention.stats()
    kv_util = runner.stats.kv_util_samples
    kv_map = runner.stats.mapped_over_reserved

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tk = torch.tensor([tokens], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        tokens = int(tk.item())

    # ---- roofline of the dominant kernel (prefill attention), from events recorded inside the timed region ----
    n_chunks = math.ceil(prefill / (a.chunk or prefill))
    launches_pf = a.steps * a.batch * n_chunks * L
    flops_total = 0.0
    c = 0
    for i in range(n_chunks):
        n = min(a.chunk or prefill, prefill - c)
        flops_total += 4.0 * Hq * D * (n * c + n * (n + 1) / 2)
        c += n
    flops_per_launch = flops_total / n_chunks
    pf_ms = op_ms.get("attn_prefill", 0.0) / max(1, launches_pf)
    pf_tflops = flops_per_launch / (pf_ms * 1e-3) / 1e12 if pf_ms > 0 else 0.0
    # decode: per launch (one layer, one iteration, batch B at ~ctx): KV read once + q,o
    dec_iters = decode - 1
    launches_dc = a.steps * dec_iters * L
    mean_len = prefill + 1 + (dec_iters - 1) / 2.0
    bytes_dc = a.batch * (2 * mean_len * Hkv * D * 2) + a.batch * Hq * D * 2 * 2
    dc_ms = op_ms.get("attn_decode", 0.0) / max(1, launches_dc)
    dc_gbs = bytes_dc / (dc_ms * 1e-3) / 1e9 if dc_ms > 0 else 0.0

    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, read side doubled per the gfx950 note in MI355X_MICROARCH.md); null for shapes that were not profiled
    is_default = (a.model == "yi-6b" and a.ctx == 32768 and a.batch == 16 and a.pd_ratio == 500.0 and not a.chunk and a.tp == 1
                  and not a.layers and a.page_size == 2 << 20)
    traffic_pf = traffic_dc = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if a.model == "yi-6b" and a.ctx == 32768 and not a.chunk and a.tp == 1:
            traffic_pf = tj["prefill_yi6b_n32702"]["hbm_bytes_per_launch"]
        if a.model == "yi-6b" and a.ctx == 32768 and a.batch == 16 and a.tp == 1:
            traffic_dc = tj["decode_yi6b_b16_32k"]["hbm_bytes_per_launch"]
    except Exception:
        pass

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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: " if is_default else "custom (NOT the bench line; one rank of TP=%d): " % a.tp) +
                            "%s fa_vattn_2mb static trace @ %d ctx, P:D=%g (%d prefill + %d decode tokens/request), "
                            "vLLM scheduler%s, one step = one max_batch_size=%d wave of requests over all %d layers, page %d KiB, async mapping; "
                            "attention+KV hot path only (transformer GEMMs out of scope, q/k/v synthetic)"
                            % (a.model + (" TP=1" if is_default else ""), a.ctx, a.pd_ratio, prefill, decode, "" if not a.chunk else " chunk=%d" % a.chunk, a.batch, L, a.page_size >> 10),
                "requests_per_step": a.batch, "layers": L, "hq": Hq, "hkv": Hkv, "head_dim": D,
                "parallelism": "replicas x%d (no collective on the path)" % world if world > 1 else "single GPU",
            },
            "roofline": {"kernel": "prefill_kernel (causal prefill attention)", "bound": "mfma", "achieved": round(pf_tflops, 2),
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(pf_tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic_pf,
                         "ms_per_launch": round(pf_ms, 4), "flops_per_launch": flops_per_launch},
            "roofline_decode": {"kernel": "decode_kernel+combine (split-KV decode, batch %d)" % a.batch, "bound": "hbm",
                                "achieved": round(dc_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(dc_gbs / HBM_PEAK_GBS, 4), "traffic": traffic_dc, "ms_per_launch": round(dc_ms, 4),
                                "bytes_per_launch": bytes_dc},
            "kv_hbm_util": {"live_over_mapped_mean": round(sum(kv_util) / max(1, len(kv_util)), 4),
                            "live_over_mapped_min": round(min(kv_util), 4) if kv_util else None,
                            "mapped_over_pool_max": round(max(kv_map), 4) if kv_map else None},
            "page_mapping": {"map_calls": vm1["map_calls"] - vm0["map_calls"], "unmap_calls": vm1["unmap_calls"] - vm0["unmap_calls"],
                             "sync_ms": round((vm1["sync_ns"] - vm0["sync_ns"]) / 1e6, 3),
                             "async_ms": round((vm1["async_ns"] - vm0["async_ns"]) / 1e6, 3),
                             "join_wait_ms": round((vm1["join_wait_ns"] - vm0["join_wait_ns"]) / 1e6, 3)},
            "op_ms": {k: round(v, 2) for k, v in op_ms.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, dtype)
        print(json.dumps(out), flush=True)
    runner.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
