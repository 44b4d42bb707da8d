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
  N = 8  configs[4]: Llama-3-70B TP = 8 (8 / 1 heads per rank, 80 layers), dynamic arxiv trace: one step = a closed-loop replay of ALL
         256 requests of the reference's length recipe (tests/golden/c3_arxiv_lengths_256.json; decode lengths capped at 768 tokens,
         stated in config.workload), max_batch_size 256, pool = 0.9 x HBM - 12 GiB per rank, megacache layout with 8 MiB pages (the
         configured 256 KiB pages need one hipMemCreate handle per page: O(live handles)).
  For N > 1 every rank processes the SAME requests with its head shard (sarathi/config.py:139-167); there is no data-path
  collective.  Inside the timed loop every iteration does the control-plane exchange of the reference engine over RCCL:
  all-reduce MIN of num_free_kvblocks() (base_llm_engine.py:381-390) and an all-gather of a fingerprint of the page-manager state,
  checked at the end of the step (identical page decisions on every rank).  value = tokens of ONE request stream / max-rank time.

Output: ONE line on stdout, the bench line of the contract, compact: its `roofline` object holds the dominant kernel AND, under `other`,
both kernels' rooflines plus the decode / prefill fractions of the dynamic legs; `legs` is a digest of the legs outside the timed
region.  Every leg in full goes to stderr as one {"details": {...}} line.

Every line (every N, static or dynamic) carries:
  `roofline`          the kernel with the largest summed time on this rank: ALGORITHMIC work of the timed launches / their summed
                      duration, from HIP events on the launch stream inside the timed region (every launch for the static
                      workloads, every 8th for the dynamic replay, whose launches are ragged: work is accounted per launch);
  `roofline_prefill` / `roofline_decode`  the same for each of the two attention kernels;
  `cpu_baseline`      the CPU oracle (kind "port": oracle/attn.py, the reference kernel's numerics in torch fp32) timed on a bounded
                      sample of THIS workload's per-rank shapes on the host cores, scaled to the whole job by the counted
                      (query row, visible key) pairs of the timed steps.
  `clock_mhz_mean` / `power_w_mean`  shader clock and board power sampled over the timed region by a side process
                      (vattention_amd/telemetry.py): a power-capped MI355X runs this kernel between ~1.5 and 2.4 GHz, boxes differ.
  `roofline.frac_at_clock` / `.mfma_busy_frac` / `.kernel_us_per_launch` (round 6)  MFMA utilisation three ways — of the 2.5 PF spec peak (`frac`), of the
                      peak at the sustained clock, and the matrix-pipe duty of the committed SQ pass — and the kernels' own durations from the
                      committed rocprofv3 trace beside the event times (profiles/rNN_traffic.json, tools/traffic_json.py).
  `roofline.other.power_ceiling` (round 6, N = 1)  what the board's POWER budget leaves of the MFMA peak on random operands, measured on this box
                      right after the timed region by tools/power_ceiling_probe (whole-chip instruction streams, no data flow, ~0.7 s each):
                      `mfma_only_tflops` (back-to-back v_mfma_f32_32x32x16_f16) and `tile_step_stream_tflops` (the same with the prefill tile
                      step's VALU mix and LDS fragment reads per MFMA); `prefill_over_mfma_only` / `prefill_over_tile_step_stream` = the prefill
                      kernel's achieved rate over them.  The part clocks to its power cap: these, not 2.5 PF, bound a chip-filling kernel.
  `roofline.other.hbm_ceiling` (round 6, N = 1)  the same probe's read-only HBM stream (4 GiB, 16-byte loads, nothing written): `read_stream_gbs`, its
                      fraction of the 8 TB/s spec peak, and `decode_over_read_stream` = the decode kernel's achieved rate over it.
  `legs.scale_series` ONE fixed workload at every N — Yi-34B (56 / 8 heads / N, 60 layers), one 131 072-token request in 16 k chunks — so that tokens/s
                      across the N = 1 / 2 / 4 / 8 lines is a strong-scaling series (the lines' `value`s follow BASELINE.json's per-N configs);
                      N > 1 adds `scaling_reference`: the same step once more WITHOUT the control-plane exchange (= --rank-of N on one GPU).
N = 1 adds, outside the timed region: `cold_wave`, `full_trace_50req`, `dynamic` (configs[2] shape, closed loop, time-weighted KV
utilisation), `dynamic_tp8_rank` (the TP8 rank shape: 256 sequences resident at full depth, deferred reclamation on / off),
`c4_rank_share_128k` (ONE rank's share of configs[3]'s 128 k request: Yi-34B TP2 heads, 16 k chunks — the metric's 128 k half on one
GPU), `hybrid_sarathi` (Sarathi-scheduled hybrid batches: serial order vs prefill || decode on two streams), `open_loop` (Poisson
arrivals at qps = 6 on a virtual clock), `capacity` (grow until the reference's OOM error).  `cpu_baseline.bookkeeping`: the
reference's own allocator (oracle/_ref) beside this manager on one engine-shaped call sequence, driver calls free.
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
    8: dict(model="llama-3-70b", tp=8, ctx=32768, pd=0.0, batch=256, chunk=0, page=8 << 20, mode="dynamic", requests=256, backend="fa_vattn_megacache",
            decode_cap=768,
            label="configs[4]: llama-3-70b TP=8 (8/1 heads per rank, 80 layers) dynamic arxiv trace, closed loop; one step = ALL 256 requests of "
                  "the reference's length recipe (decode lengths capped at 768 tokens: a handful of the 256 run to 6 k tokens and would leave "
                  "thousands of batch-1 iterations at the end of every step), max_batch_size 256, pool = 0.9 x HBM - 12 GiB per rank, "
                  "megacache layout with 8 MiB pages (stands in for 256 KiB pages)"),
}
# time every k-th launch of each attention operation.  The stride of the replay legs is COPRIME with every model's layer count: a replay
# issues one prefill launch per layer and iteration, so a stride of 8 on 80 (or 32) layers samples layer 0 in every iteration and seven
# layers never — and layer 0's event pair opens on an idle GPU, before the host has built and uploaded the iteration's plan, so it reads
# host latency + kernel.  Rounds 4-5 (stride 8) therefore under-read the replay legs' prefill fractions by the planner's host time x 10 / 8
# (profiles/r05_timer_stride.txt: same box, same library: stride 8 vs 7); whole-leg tokens/s never depended on it.
TIMER_EVERY = {"static": 1, "dynamic": 7}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dynamic", action="store_true", help="skip the legs outside the timed region (N = 1)")
    ap.add_argument("--no-power-ceiling", action="store_true", help="skip tools/power_ceiling_probe after the timed region (N = 1)")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; makes the number INVALID)")
    ap.add_argument("--ctx", type=int, default=0, help="override context length (debug only; makes the number INVALID)")
    ap.add_argument("--requests", type=int, default=0, help="override requests per step (debug only; makes the number INVALID)")
    ap.add_argument("--rank-of", type=int, default=0, help="debug: run ONE rank's share of the --gpus N workload of this value on a single GPU, "
                    "without the collectives (checks the tensor-parallel workloads where only one GPU is visible; NOT a bench line)")
    ap.add_argument("--leg", default="", help="run ONLY one of the legs outside the timed region (dynamic, dynamic_tp8_rank, capacity) and print it: "
                    "what tools/prof_round.sh profiles (NOT a bench line)")
    ap.add_argument("--timer-every", type=int, default=0, help="A/B: event-pair stride of the replay legs (default 7; rounds 4-5 used 8); NOT the bench line")
    ap.add_argument("--per-piece-prefill", action="store_true", help="A/B: NO persistent workgroups anywhere (prefill64p_kernel off); NOT the bench line")
    ap.add_argument("--persistent-prefill", action="store_true", help="A/B: EVERY prefill work list through persistent workgroups; NOT the bench line")
    ap.add_argument("--qps", type=float, default=0.0, help="run ONLY the open-loop replay (Poisson arrivals, reference recipe) at this rate and print it")
    return ap.parse_args()


def valid_or_dry(a) -> bool:
    """the N > 1 extras also run in the reduced dry runs (--layers / --requests over the gloo hook), not under the A/B switches"""
    return not (a.rank_of or a.per_piece_prefill or a.persistent_prefill or a.timer_every or a.ctx)


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


def cpu_rates(dtype, Hq, Hkv, D, keys, budget_head_pairs=0.42e9) -> dict:
    """The CPU oracle (oracle/attn.py, math='f32' = the reference kernel's numerics) on this workload's per-rank head shape:
    (a) prefill form — the LAST `rows` query rows of a `keys`-token causal prompt (rows chosen so that the sample is about
    budget_head_pairs (query row, key, query head) triples: 10-15 s on this class of host); (b) decode form — one step of two
    sequences at `keys` context.  Returns head-pairs per second for both forms."""
    import torch
    from oracle.attn import flash_attn_with_kvcache_ref
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    n = keys
    rows = int(max(32, min(512, budget_head_pairs / (n * Hq))))
    k = torch.randn(1, n + 8, Hkv, D).to(dtype)
    v = torch.randn(1, n + 8, Hkv, D).to(dtype)
    q = torch.randn(1, rows, Hq, D).to(dtype)
    t0 = time.perf_counter()
    flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=n, causal=True, math="f32")      # rows [n - rows, n) over keys [0, n)
    t_blk = time.perf_counter() - t0
    pairs_blk = sum(i + 1 for i in range(n - rows, n)) * Hq
    kd = torch.randn(2, n + 8, Hkv, D).to(dtype)
    vd = torch.randn(2, n + 8, Hkv, D).to(dtype)
    qd = torch.randn(2, 1, Hq, D).to(dtype)
    t0 = time.perf_counter()
    flash_attn_with_kvcache_ref(qd, kd, vd, cache_seqlens=torch.tensor([n, n + 4], dtype=torch.int32), causal=True, math="f32")
    t_dec = time.perf_counter() - t0
    pairs_dec = (2 * n + 4 + 2) * Hq
    return {"cores": cores, "rows": rows, "keys": n, "t_prefill_sample": t_blk, "t_decode_sample": t_dec,
            "prefill_head_pairs_per_s": pairs_blk / t_blk, "decode_head_pairs_per_s": pairs_dec / t_dec}


def cpu_baseline(dtype, L, Hq, Hkv, D, keys, world, tokens, prefill_pairs, decode_pairs, what) -> dict:
    """tokens/s the CPU oracle would reach on the WHOLE job of the timed steps: all `world` head shards, L layers, the (query row,
    visible key) pairs the replay counted (replay.ReplayStats.prefill_pairs / decode_pairs, per layer and per query head), at the
    head-pair rates measured on this workload's per-rank shape."""
    r = cpu_rates(dtype, Hq, Hkv, D, keys)
    t_job = world * L * Hq * (prefill_pairs / r["prefill_head_pairs_per_s"] + decode_pairs / r["decode_head_pairs_per_s"])
    return {"value": round(tokens / t_job, 3), "unit": "tokens/s", "cores": r["cores"], "kind": "port",
            "sample": "CPU oracle (oracle/attn.py, torch fp32 math on fp16 inputs, %d threads) on %s's per-rank shape (%d/%d heads, d %d): last "
                      "%d query rows of a %d-token causal prompt (%.2f s -> %.3g head-pairs/s) and one decode step of 2 sequences at %d "
                      "keys (%.3f s -> %.3g head-pairs/s); whole job = %d shard(s) x %d layers x %d heads x (%.4g prefill + %.4g decode "
                      "pairs of the timed steps) = %.0f s of CPU time for %d tokens" % (
                          r["cores"], what, Hq, Hkv, D, r["rows"], r["keys"], r["t_prefill_sample"], r["prefill_head_pairs_per_s"], r["keys"],
                          r["t_decode_sample"], r["decode_head_pairs_per_s"], world, L, Hq, prefill_pairs, decode_pairs, t_job, tokens)}


def bookkeeping_baseline() -> dict:
    """SURVEY §8(d), bookkeeping half: microseconds per step_async of the REFERENCE's allocator (oracle/_ref = vattention.cu compiled
    against a fake CUDA driver, kind "reference") and of this package's page manager (fake physical backend), same concrete call
    sequence of an engine-shaped trace at configs[1]'s geometry, all driver calls free, one host thread.  ~1 s of CPU."""
    try:
        from tools import pagemgr_steps_bench as B
        m = B.measure(iters=300)
        ref, here = m.get("reference_us_per_step_async"), m.get("this_manager_inline_us_per_step_async")
        m.update({"kind": "reference" if ref else "port", "cores": 1, "unit": "us per step_async (driver calls free)",
                  "reference_over_this": round(ref / here, 2) if ref and here else None})
        return m
    except Exception as e:      # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def measured_ceilings(prefill_tflops, decode_gbs, seconds=0.7):
    """(roofline.other.power_ceiling, roofline.other.hbm_ceiling) from ONE run of tools/power_ceiling_probe --quick"""
    pc = power_ceiling(prefill_tflops, seconds)
    h = pc.pop("hbm_read_stream_gbs", None)
    if not h:
        return pc, ({"error": pc["error"]} if "error" in pc else {"error": "the probe reported no HBM stream"})
    hc = {"read_stream_gbs": h, "read_stream_frac_of_peak": round(h / 8000.0, 4),
          "what": "read-only stream over 4 GiB (16x the MALL): 1 024 x 1 024 threads, four 16-byte loads in flight per lane, nothing written"}
    if decode_gbs:
        hc["decode_over_read_stream"] = round(decode_gbs / h, 4)
    return pc, hc


def power_ceiling(prefill_tflops, seconds=0.7) -> dict:
    """roofline.other.power_ceiling: tools/power_ceiling_probe --quick on the (now idle) GPU — the MFMA rate the board sustains under its power
    cap on pseudo-random operands, for back-to-back MFMAs and for the prefill tile step's instruction mix — and the prefill kernel's rate over them."""
    exe = os.path.join(ROOT, "tools", "power_ceiling_probe")
    if not os.path.exists(exe):
        return {"error": "tools/power_ceiling_probe is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    try:
        r = subprocess.run([exe, "--quick", str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001  (the bench line is printed either way)
        return {"error": "%s: %s" % (type(e).__name__, e)}
    d["what"] = ("whole-chip streams of v_mfma_f32_32x32x16_f16 on pseudo-random operands, one wave per SIMD, no data flow: MFMAs only / with the prefill "
                 "tile step's VALU mix and LDS fragment reads; the board clocks to its power cap")
    d["mfma_only_frac_of_peak"] = round(d["mfma_only_tflops"] / 2500.0, 4)
    d["tile_step_stream_frac_of_peak"] = round(d["tile_step_stream_tflops"] / 2500.0, 4)
    if prefill_tflops:
        d["prefill_over_mfma_only"] = round(prefill_tflops / d["mfma_only_tflops"], 4)
        d["prefill_over_tile_step_stream"] = round(prefill_tflops / d["tile_step_stream_tflops"], 4)
    return d


def rooflines(detail: dict, traffic=None) -> dict:
    """roofline objects from the op-timer records (attention/timers.py): achieved = algorithmic work of the TIMED launches / their
    summed duration; the dominant kernel is the one with the larger estimated total time."""
    out = {}
    spec = {"attn_prefill": ("prefill", "prefill attention (causal chunk(s) against the KV prefix; prefill64_kernel / prefill_kernel + combine), per rank",
                             "mfma", MFMA_PEAK_TFLOPS, "TFLOP/s", 1e12, "flops"),
            "attn_decode": ("decode", "decode_stream_kernel + decode_stream_combine_kernel (split-KV decode planned on the device from cache_seqlens, in-kernel append; "
                            "decode_kernel + combine_kernel for a single sequence), per rank", "hbm", HBM_PEAK_GBS, "GB/s", 1e9, "bytes")}
    est = {}
    for op, (short, name, bound, peak, unit, div, wname) in spec.items():
        d = detail.get(op)
        if not d or not d["timed"] or d["ms"] <= 0:
            continue
        ach = d["work"] / (d["ms"] * 1e-3) / div
        est[short] = d["ms"] * d["n"] / d["timed"]
        out["roofline_" + short] = {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                                    "frac": round(ach / peak, 4), "traffic": (traffic or {}).get(short),
                                    "ms_per_launch": round(d["ms"] / d["timed"], 4), wname + "_per_launch": d["work"] / d["timed"],
                                    "launches": d["n"], "launches_timed": d["timed"], "est_total_ms": round(est[short], 1)}
    if est:
        dom = max(est, key=est.get)
        out["roofline"] = dict(out["roofline_" + dom], dominant_of=sorted(est))
    return out


def mfma_views(roofs: dict, clock_mhz, traffic) -> None:
    """adds `frac_at_clock` (achieved / (2.5 PF x clock / 2400 MHz)) and `mfma_busy_frac` (from the committed PMC pass) to the MFMA-bound rooflines"""
    for key in ("roofline", "roofline_prefill"):
        r = roofs.get(key)
        if r and r.get("bound") == "mfma":
            if clock_mhz:
                r["frac_at_clock"] = round(r["achieved"] / (MFMA_PEAK_TFLOPS * clock_mhz / 2400.0), 4)
                r["clock_mhz_mean"] = clock_mhz
            if traffic and traffic.get("mfma_busy_frac") is not None:
                r["mfma_busy_frac"] = traffic["mfma_busy_frac"]
                r["mfma_busy_source"] = traffic["source"] + " (separate --pmc pass over the same launch, not this run)"
    # `kernel_us_per_launch`: the kernels' own mean duration (rocprofv3 kernel trace of the same workload, committed) beside the event time
    # `ms_per_launch`, which also holds the launch boundaries — a 0.697 vs 0.717 decode reading can then be told apart as gap or kernel
    for key, short in (("roofline_prefill", "prefill"), ("roofline_decode", "decode")):
        r = roofs.get(key)
        ku = ((traffic or {}).get("kernel_us") or {}).get(short)
        if r and ku:
            r["kernel_us_per_launch"] = ku
            r["kernel_us_source"] = traffic["source"] + " (rocprofv3 --kernel-trace --stats of this workload, not this run; decode = stream kernel + merge kernel)"
            if roofs.get("roofline", {}).get("kernel") == r.get("kernel"):
                roofs["roofline"]["kernel_us_per_launch"] = ku


_REAL_STDOUT = None


def _own_stdout():
    """stdout carries the bench line and nothing else: whatever the libraries print on fd 1 (the page manager dumps its state there before
    an OOM error, as the reference does, vattention.cu:294-295 — the capacity leg provokes one) is sent to stderr from here on."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line: str):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    _own_stdout()
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
    from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers
    from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig
    if a.timer_every > 0:
        TIMER_EVERY["dynamic"] = a.timer_every
    if a.per_piece_prefill or a.persistent_prefill:
        from vattention_amd import flash_attn as _FA
        _FA.PERSISTENT = "never" if a.per_piece_prefill else "always"

    if a.rank_of and (world != 1 or a.rank_of not in WORKLOADS):
        raise SystemExit("--rank-of needs --gpus 1 and one of %s" % sorted(WORKLOADS))
    shard_of = a.rank_of or world                            # tensor-parallel degree of the workload this process runs a rank of
    w = dict(WORKLOADS[shard_of])
    if a.ctx:
        w["ctx"] = a.ctx
    if a.requests:
        w["requests"] = a.requests
    dtype = torch.float16                                    # benchmark_runner.py:81
    valid = not (a.layers or a.ctx or a.rank_of or a.requests or a.per_piece_prefill or a.persistent_prefill or a.timer_every)

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
    lengths256 = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
    cap = lambda ls, c: [[pre, min(dec, c)] for pre, dec in ls]

    if a.leg:       # one leg on its own (profiling), not a bench line
        if a.leg == "dynamic":
            res = dynamic_leg(make_runner, mem_for_kv, lengths256, "llama-3-8b", 1, "configs[2] shape: llama-3-8b TP=1, 32 layers", None, dtype, False)
        elif a.leg == "dynamic_tp8_rank":
            res = dynamic_leg(make_runner, mem_for_kv, cap(lengths256, 768), "llama-3-70b", 8, "one TP=8 rank of configs[4]", None, dtype, False)
        elif a.leg == "capacity":
            res = capacity_leg(dev, mem_for_kv)
        elif a.leg == "c4_rank_share_128k":
            res = c4_rank_share_leg(make_runner, mem_for_kv)
        elif a.leg == "hybrid_sarathi":
            res = hybrid_sarathi_leg(make_runner, mem_for_kv)
        elif a.leg == "hybrid_sarathi_1k_chunks":
            res = hybrid_sarathi_leg(make_runner, mem_for_kv, 1024)
        elif a.leg == "hybrid_sarathi_512_chunks":
            res = hybrid_sarathi_leg(make_runner, mem_for_kv, 512)
        elif a.leg == "scale_series":
            res = scale_series_leg(make_runner, mem_for_kv, 1)
        else:
            raise SystemExit("--leg must be dynamic, dynamic_tp8_rank, c4_rank_share_128k, scale_series, hybrid_sarathi or capacity")
        _emit(json.dumps({a.leg: res}))
        return
    if a.qps:       # stand-alone open-loop replay (not a bench line)
        _emit(json.dumps(open_loop_leg(make_runner, mem_for_kv, lengths256, a.qps, a.requests or 256)))
        return

    runner = make_runner(w["model"], w["tp"], w["ctx"], w["page"], w["batch"], w["backend"], mem_for_kv, a.layers)
    pool_ready_s = round(vattention.pool_ready_seconds, 3)      # reserve_physical_pages waited this long for the handles created ahead of demand
    Hq, Hkv, D, L = runner.Hq, runner.Hkv, runner.D, runner.L
    lengths = cap(lengths256, w["decode_cap"]) if w["mode"] == "dynamic" and w.get("decode_cap") else lengths256

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

    pairs = {"pf": 0.0, "dc": 0.0}

    def one_step():
        runner.stats.__init__()
        if w["mode"] == "static":
            runner.run_static_trace(w["requests"], w["ctx"], w["pd"], w["chunk"] or None)
        else:
            runner.run_dynamic_trace(w["requests"], lengths=lengths)
        pairs["pf"] += runner.stats.prefill_pairs
        pairs["dc"] += runner.stats.decode_pairs
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
    pairs["pf"] = pairs["dc"] = 0.0
    enable_op_timers(True, every=TIMER_EVERY[w["mode"]])      # HIP events on the launch stream, inside the timed region
    from vattention_amd.telemetry import Sampler
    sampler = Sampler(local, 0.1).__enter__() if rank == 0 else None      # a side process: shader clock and board power
    barrier()
    wall0 = time.time()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(a.steps):
        tokens += one_step()
    barrier()
    dt = time.perf_counter() - t0
    telemetry = sampler.window(wall0, time.time()) if sampler is not None else None
    detail = drain_op_timers_detail()
    enable_op_timers(False)
    vm1 = vattention.stats()
    kv_util = list(runner.stats.kv_util_samples)
    kv_map = list(runner.stats.mapped_over_reserved)
    peak_running = None
    if cold is not None:
        cold["warm_step_ms"] = round(dt * 1e3 / a.steps, 1)

    tp_check = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tp_check = {"iterations_with_control_plane_exchange": ctl["iters"], "min_free_kvblocks_over_ranks": int(ctl["log"].item()),
                    "identical_page_decisions_on_all_ranks": bool(int(ctl["ok"].item()))}
        if not tp_check["identical_page_decisions_on_all_ranks"]:
            raise SystemExit("tensor-parallel ranks diverged in their page-manager state")

    # ---- roofline of the attention kernels, from events recorded inside the timed region ----
    traffic = None
    try:       # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 per the gfx950 note); configs[1] only
        # (PMC counters cannot be collected inside the timed run: they come from the newest committed separate pass over the SAME two
        # launches, tools/prof_round.sh -> tools/traffic_json.py; `traffic_source` names the file)
        for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            pth = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pth) and world == 1 and valid:
                tj = json.load(open(pth))
                traffic = {"prefill": tj["prefill_yi6b_n32702"]["hbm_bytes_per_launch"], "decode": tj["decode_yi6b_b16_32k"]["hbm_bytes_per_launch"],
                           "mfma_busy_frac": tj["prefill_yi6b_n32702"].get("mfma_busy_frac"),
                           "kernel_us": {"prefill": tj["prefill_yi6b_n32702"].get("kernel_us"), "decode": tj["decode_yi6b_b16_32k"].get("kernel_us")},
                           "source": "profiles/" + name}
                break
    except Exception:
        traffic = None
    roofs = rooflines(detail, traffic)
    for key in ("roofline", "roofline_prefill", "roofline_decode"):
        if key in roofs and traffic:
            roofs[key]["traffic_source"] = traffic["source"] + " (separate --pmc pass over the same launch, not this run)"
    # MFMA utilisation the three ways a reader may mean it: `frac` of the 2.5 PF spec peak (2.4 GHz), `frac_at_clock`
    # of the peak at the shader clock this run sustained (a power-capped part runs this kernel at 1.7-1.9 GHz), and `mfma_busy_frac` =
    # matrix-pipe duty from the committed PMC pass over the same launch (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) over
    # GRBM_GUI_ACTIVE / 8 XCDs; like `traffic`, counters cannot be collected inside the timed run)
    mfma_views(roofs, (telemetry or {}).get("clock_mhz_mean"), traffic)
    if "roofline" in roofs:
        # how the launches were timed: per-launch HIP event pairs, except decode-only iterations: ONE pair around the
        # iteration's L back-to-back decode launches (an event pair costs microseconds of its own on 60-150 us launches)
        roofs["roofline"]["timing"] = ("HIP events on the launch stream; prefill per launch, decode-only iterations one pair per %d launches; replay legs (other.dynamic_*): "
                                       "every %d-th launch / iteration, a stride coprime with the layer count" % (L, TIMER_EVERY["dynamic"]))
    if dist is not None and "roofline" in roofs:      # the same kernel's mean launch time on every rank
        for key in ("roofline_prefill", "roofline_decode"):
            mine = roofs.get(key, {}).get("ms_per_launch", 0.0)
            g = torch.tensor([mine], dtype=torch.float64, device=red_dev)
            gl = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(gl, g)
            if key in roofs:
                roofs[key]["ms_per_launch_by_rank"] = [round(float(x.item()), 4) for x in gl]
    op_ms = {k: round(v["ms"] * (v["n"] / v["timed"] if v["timed"] else 0.0), 2) for k, v in detail.items()}

    # ---- N > 1, outside the timed region: (a) `scaling_reference` — the SAME step once more on every rank with the control-plane
    # exchange switched off: one rank's share of the work on its own, what `bench.py --rank-of N` measures on a single GPU; value /
    # that = what the exchange and the ranks' skew cost.  (b) `scale_series` — the fixed Yi-34B 128 k request at this N (the main
    # workload itself at N = 2 / 4; run here at N = 8) ----
    scaling_reference = series = None
    if dist is not None and valid_or_dry(a):
        hook_saved, runner.iter_hook = runner.iter_hook, None
        pairs_saved = dict(pairs)                        # (the cpu_baseline scales by the pairs of the TIMED steps)
        barrier()
        t1 = time.perf_counter()
        tk_ref = one_step()
        torch.cuda.synchronize()
        dt_ref = time.perf_counter() - t1
        runner.iter_hook = hook_saved
        pairs.update(pairs_saved)
        tr = torch.tensor([dt_ref], dtype=torch.float64, device=red_dev)
        gl = [torch.zeros_like(tr) for _ in range(world)]
        dist.all_gather(gl, tr)
        per_rank = [float(x.item()) for x in gl]
        scaling_reference = {"what": "one more step on every rank WITHOUT the control-plane exchange (a rank's share on its own, = bench.py --rank-of %d on one GPU)" % world,
                             "tokens_per_s_slowest_rank": round(tk_ref / max(per_rank), 2), "seconds_by_rank": [round(x, 4) for x in per_rank],
                             "value_over_reference": round((tokens / dt) / (tk_ref / max(per_rank)), 4)}
        if w["model"] == "yi-34b" and w["ctx"] == 131072 and not a.requests:
            series = {"n_gpus": world, "tokens_per_s": round(tokens / dt, 2), "seconds": round(dt / a.steps, 3), "same_as": "value (the main workload of this N is the series' workload)"}
        else:
            def mx(x):
                t = torch.tensor([x], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            runner.close()
            runner = None
            try:
                series = scale_series_leg(lambda *args: make_runner(*args, layers=a.layers), mem_for_kv, world, hook_saved, mx)
            except Exception as e:      # noqa: BLE001  (every rank fails alike: shapes and memory are the same)
                series = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- N = 1 extras, outside the timed region ----
    extras = {}
    if world == 1 and valid and not a.no_dynamic:
        torch.cuda.synchronize()
        try:
            runner.stats.__init__()
            t1 = time.perf_counter()
            runner.run_static_trace(50, w["ctx"], w["pd"], None)        # the whole 50-request trace: 16 + 16 + 16 + 2
            torch.cuda.synchronize()
            dt_f = time.perf_counter() - t1
            tk_f = runner.stats.prefill_tokens + runner.stats.decode_tokens
            extras["full_trace_50req"] = {"requests": 50, "tokens": tk_f, "seconds": round(dt_f, 3), "tokens_per_s": round(tk_f / dt_f, 1)}
        except Exception as e:      # noqa: BLE001
            extras["full_trace_50req"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            runner.close()
        except Exception:      # noqa: BLE001
            pass
        runner = None
        # every leg on its own: a failure in one of them is reported in its place and never costs the bench line
        def leg(name, fn, *args):
            try:
                extras[name] = fn(*args)
            except Exception as e:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    vattention.cleanup()
                except Exception:      # noqa: BLE001
                    pass
                torch.cuda.synchronize()

        leg("dynamic", dynamic_leg, make_runner, mem_for_kv, lengths256, "llama-3-8b", 1, "configs[2] shape: llama-3-8b TP=1, 32 layers",
            None, dtype, not a.no_cpu_baseline)
        leg("dynamic_tp8_rank", dynamic_leg, make_runner, mem_for_kv, cap(lengths256, 768), "llama-3-70b", 8,
            "one TP=8 rank of configs[4]: llama-3-70b, 8/1 heads, 80 layers (40 KB of KV per token: 256 sequences fit at full depth); decode "
            "lengths capped at 768", True, dtype, False)
        leg("c4_rank_share_128k", c4_rank_share_leg, make_runner, mem_for_kv)
        leg("scale_series", scale_series_leg, make_runner, mem_for_kv, 1)
        leg("hybrid_sarathi", hybrid_sarathi_leg, make_runner, mem_for_kv)
        # the regime the reference's POD kernel targets: chunks that leave CUs free (1 024 tokens x 32 heads = 128 workgroups on 256 CUs)
        leg("hybrid_sarathi_1k_chunks", hybrid_sarathi_leg, make_runner, mem_for_kv, 1024)
        leg("open_loop", open_loop_leg, make_runner, mem_for_kv, lengths256, 6.0, 256)
        leg("capacity", capacity_leg, dev, mem_for_kv)

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
            "op_ms": op_ms,
            "pool_ready_s": pool_ready_s,
        }
        if telemetry is not None:
            out["clock_mhz_mean"] = telemetry.get("clock_mhz_mean")
            out["power_w_mean"] = telemetry.get("power_w_mean")
            out["telemetry"] = telemetry
        out.update(roofs)
        if "roofline" in out:
            # BOTH kernels' rooflines inside the object the driver keeps (`roofline`): the north_star's two bars — decode vs the HBM
            # roofline, prefill vs the MFMA roofline — are read from the same place whichever kernel dominates the step; plus the two
            # fractions of each dynamic leg (ragged launches, accounted per launch)
            pick = lambda r: None if not r else {k: r.get(k) for k in ("bound", "frac", "achieved", "peak", "unit", "ms_per_launch", "bytes_per_launch", "flops_per_launch",
                                                                       "traffic", "launches_timed", "frac_at_clock", "mfma_busy_frac", "kernel_us_per_launch") if r.get(k) is not None}
            other = {"decode": pick(roofs.get("roofline_decode")), "prefill": pick(roofs.get("roofline_prefill"))}
            # the dynamic legs: the replay's second pass (warm pool: every handle exists — the serving steady state) and its first
            # (cold pool: the mapper thread is still creating the pool's handles under the first iterations)
            for leg in ("dynamic", "dynamic_tp8_rank"):
                e = extras.get(leg) or {}
                for kern in ("decode", "prefill"):
                    fr_w = ((e.get("warm_pool_pass") or {}).get("roofline_" + kern) or {}).get("frac")
                    fr_c = (e.get("roofline_" + kern) or {}).get("frac")
                    if fr_w is not None:
                        other["%s_%s_frac" % (leg, kern)] = fr_w
                    if fr_c is not None:
                        other["%s_%s_frac_cold_pool" % (leg, kern)] = fr_c
            e128 = extras.get("c4_rank_share_128k") or {}
            if e128.get("prefill_frac") is not None:
                other["c4_rank_share_128k_prefill_frac"] = e128.get("prefill_frac")
                other["c4_rank_share_128k_decode_frac"] = e128.get("decode_frac")
            out["roofline"]["other"] = other
        if cold:
            out["cold_wave"] = cold
        if tp_check:
            out["tensor_parallel"] = tp_check
        if scaling_reference:
            out["scaling_reference"] = scaling_reference
        if series:
            out["_series"] = series
        out["_extras"] = extras
    if runner is not None:
        runner.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not a.no_power_ceiling and not a.leg and isinstance((out.get("roofline") or {}).get("other"), dict):
            pf, dc = out["roofline"]["other"].get("prefill") or {}, out["roofline"]["other"].get("decode") or {}
            pc, hc = measured_ceilings(pf.get("achieved") if pf.get("unit") == "TFLOP/s" else None, dc.get("achieved") if dc.get("unit") == "GB/s" else None)
            out["roofline"]["other"]["power_ceiling"], out["roofline"]["other"]["hbm_ceiling"] = pc, hc
            # the dominant kernel's rate over its measured ceiling, beside `frac` (of the spec peak)
            over = pc.get("prefill_over_mfma_only") if out["roofline"].get("bound") == "mfma" else hc.get("decode_over_read_stream")
            if over is not None:
                out["roofline"]["frac_of_measured_ceiling"] = over
                out["roofline"]["measured_ceiling"] = ("back-to-back MFMAs on random operands under the board's power cap (other.power_ceiling.mfma_only_tflops)"
                                                       if out["roofline"].get("bound") == "mfma" else "read-only HBM stream (other.hbm_ceiling.read_stream_gbs)")
        if not a.no_cpu_baseline:
            # after the timed region and after the ranks have parted: the oracle on this workload's per-rank shape, host cores only
            keys = min(w["ctx"], 32768) if w["mode"] == "dynamic" else w["ctx"] - math.ceil(w["ctx"] / (1 + w["pd"]))
            try:
                out["cpu_baseline"] = cpu_baseline(dtype, L, Hq, Hkv, D, keys, shard_of, tokens, pairs["pf"], pairs["dc"],
                                                   "configs[%d]" % {1: 1, 2: 3, 4: 3, 8: 4}[shard_of])
            except Exception as e:      # noqa: BLE001  (the line is printed either way)
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["cpu_baseline"]["bookkeeping"] = bookkeeping_baseline()
        # stdout carries ONE line, the bench line the contract asks for — compact, so that whoever reads the tail of stdout sees a whole
        # line: the contract's fields, `roofline` (with both kernels and the dynamic legs' fractions under `other`), `cpu_baseline`, and
        # a digest of the legs.  The details (every leg in full: tens of kilobytes) go to stderr as one {"details": ...} line.
        extras = out.pop("_extras", {})
        series = out.pop("_series", None) or extras.get("scale_series")
        details = dict(extras)
        for k in ("roofline_prefill", "roofline_decode", "op_ms", "cold_wave"):
            if k in out:
                details[k] = out.pop(k)
        print(json.dumps({"details": details}), file=sys.stderr, flush=True)
        dig = {}

        def g(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        if "cold_wave" in details:
            dig["cold_wave_ms"] = details["cold_wave"].get("ms")
            dig["cold_wave_sync_share_of_map_time"] = details["cold_wave"].get("sync_share_of_map_time")
        for leg in ("dynamic", "dynamic_tp8_rank"):
            e = extras.get(leg)
            if e:
                dig[leg] = {"tokens_per_s_cold": e.get("tokens_per_s"), "tokens_per_s_warm": g(e, "warm_pool_pass", "tokens_per_s"),
                            "cold_over_warm": e.get("cold_over_warm_tokens_per_s"), "peak_concurrent_sequences": e.get("peak_concurrent_sequences"),
                            "kv_live_over_needed_at_peak": e.get("kv_live_over_needed_at_peak"),
                            "steady_state_live_over_mapped": g(e, "kv_util_time_weighted", "steady_state", "live_over_mapped"),
                            "create_ms_on_critical_path": g(e, "handle_creation", "on_critical_path_ms"), "error": e.get("error")}
        if extras.get("full_trace_50req"):
            dig["full_trace_50req_tokens_per_s"] = extras["full_trace_50req"].get("tokens_per_s")
        if extras.get("c4_rank_share_128k"):
            dig["c4_rank_share_128k"] = {k: extras["c4_rank_share_128k"].get(k) for k in ("tokens_per_s", "seconds", "prefill_frac", "decode_frac", "kv_live_over_mapped_mean", "error")}
        if extras.get("hybrid_sarathi"):
            dig["hybrid_sarathi"] = {k: extras["hybrid_sarathi"].get(k) for k in ("tokens_per_s_serial", "tokens_per_s_streams", "streams_over_serial", "hybrid_iterations",
                                                                                  "overlapped_iterations", "error")}
        if extras.get("hybrid_sarathi_1k_chunks"):
            dig["hybrid_sarathi_1k_chunks"] = {k: extras["hybrid_sarathi_1k_chunks"].get(k) for k in ("tokens_per_s_serial", "tokens_per_s_streams", "streams_over_serial",
                                                                                                      "hybrid_iterations", "overlapped_iterations", "error")}
        if extras.get("open_loop"):
            dig["open_loop_qps6"] = {k: extras["open_loop"].get(k) for k in ("request_e2e_time_normalized_p50", "request_e2e_time_normalized_p99", "sync_map_ms_per_step_p99")}
        if extras.get("capacity"):
            dig["capacity"] = {k: extras["capacity"].get(k) for k in ("mapped_over_budget", "mapped_over_hbm", "tokens_resident", "fill_seconds")}
        if series:
            # legs.scale_series: the SAME request at every N — read tokens_per_s across the N = 1 / 2 / 4 / 8 lines for a strong-scaling curve
            dig["scale_series"] = {k: series.get(k) for k in ("n_gpus", "tokens_per_s", "seconds", "prefill_frac", "decode_frac", "same_as", "error") if series.get(k) is not None}
            dig["scale_series"]["workload"] = "yi-34b (56/8 heads / N, 60 layers), ONE 131072-token request, P:D=500, 16 k chunks, 2 MiB pages: identical at every N (strong scaling)"
        if dig:
            out["legs"] = dig
        _emit(json.dumps(out))


def dynamic_leg(make_runner, mem_for_kv, lengths, model, tp, what, ab_deferred, dtype, with_cpu) -> dict:
    """A closed-loop replay of the 256-request arxiv-length recipe at FULL depth on a fresh (cold) pool: peak concurrency, internal
    fragmentation at the peak, KV utilisation weighted by GPU time (steady-state window / drain tail), mapping cost split by thread.
    Megacache layout with 8 MiB pages: 64 / 256 KiB pages would need 1-4 M hipMemCreate handles for this pool (O(live handles) on
    ROCm).  ab_deferred: run the same replay again with set_deferred_reclamation(False) and report both."""
    import torch
    from vattention_amd import vattention
    from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers

    def run(deferred: bool, warm_pass: bool = False):
        r = make_runner(model, tp, 32768, 8 << 20, 256, "fa_vattn_megacache", mem_for_kv)
        ready_s = round(vattention.pool_ready_seconds, 3)
        try:
            if not deferred:
                r.engine.disable_deferred_reclamation()
            r.time_iterations = True
            enable_op_timers(True, every=TIMER_EVERY["dynamic"])
            out = r.run_dynamic_trace(256, lengths=lengths)
            det = drain_op_timers_detail()
            enable_op_timers(False)
            out["_stats"] = (r.stats.prefill_pairs, r.stats.decode_pairs, r.L, r.Hq, r.Hkv, r.D)
            out["_pool_ready_s"] = ready_s
            out["_roof"] = rooflines(det)
            if warm_pass:
                # the SAME replay again on the now-warm pool (every handle exists, finished slots kept their pages): what the cold
                # pass pays for creating the pool's handles while it runs is the difference
                r.stats.__init__()
                enable_op_timers(True, every=TIMER_EVERY["dynamic"])
                w2 = r.run_dynamic_trace(256, lengths=lengths)
                det2 = drain_op_timers_detail()
                enable_op_timers(False)
                w2["_roof"] = rooflines(det2)
                out["_warm"] = w2
            return out
        finally:
            r.close()

    def digest(out):
        # synchronous batches = driver calls on the engine thread (maps / unmaps / set-access / creations / TLB invalidation) PLUS,
        # before an unmap, the wait for the fence of the slot that gives the page up (not mapping work; it moves to the mapper
        # thread when the look-ahead does the reclaim)
        sb = out.get("sync_breakdown") or {}
        fence_ms = float(sb.get("fence_ms", 0.0))
        sync_calls_ms = max(0.0, out["sync_map_ms"] - fence_ms)
        tot = sync_calls_ms + out["async_map_ms"]
        d = {"peak_concurrent_sequences": out["peak_running"], "tokens": out["tokens"], "seconds": round(out["seconds"], 2),
             "tokens_per_s": round(out["tokens_per_s"], 1), "iterations": out["iters"],
             "kv_live_over_needed_at_peak": out["kv_live_over_needed_at_peak"],
             "kv_live_over_mapped_mean_per_iteration": round(out["kv_live_over_mapped_mean"], 4),
             "kv_util_time_weighted": out.get("kv_util_time_weighted"),
             "external_fragmentation": out.get("external_fragmentation_max"), "external_fragmentation_samples": out.get("external_fragmentation_samples"),
             "map_calls": out["map_calls"], "unmap_calls": out["unmap_calls"],
             "handles_created": out.get("handles_created"), "create_ms": out.get("create_ms"),
             "sync_map_ms": round(sync_calls_ms, 1), "sync_fence_wait_ms": round(fence_ms, 1),
             "mapper_thread_map_ms": round(out["async_map_ms"], 1),
             "sync_share_of_map_time": round(sync_calls_ms / tot, 4) if tot else None,
             "sync_map_share_of_wall": round(out["sync_map_ms"] / 1e3 / out["seconds"], 4),
             "sync_breakdown": out.get("sync_breakdown")}
        d.update(out["_roof"])
        return d

    first = run(True, warm_pass=True)
    res = {"workload": what + "; 256 arxiv-length requests (tests/golden/c3_arxiv_lengths_256.json) closed loop, vLLM scheduler, max_batch_size 256, "
                              "megacache 8 MiB pages, pool = 0.9 x HBM - 12 GiB; first pass on a fresh pool whose handles were created inside "
                              "reserve_physical_pages (pool_ready_s), second pass warm"}
    res.update(digest(first))
    res["timing"] = ("HIP events on the launch stream; hybrid iterations: every %d-th launch of each operation (a stride coprime with the layer count: every layer is "
                     "sampled equally often), decode-only iterations: one pair around the iteration's launches, every %d-th iteration" % (TIMER_EVERY["dynamic"], TIMER_EVERY["dynamic"]))
    # the pool's handles exist before the first admission (vattention.reserve_physical_pages waits for the mapper thread: what the
    # reference's reserve commits, cudaInternal.h:45-59); "cold" below = the first pass on that fresh pool, "warm" = the second pass
    res["pool_ready_s"] = first.get("_pool_ready_s")
    warm = digest(first["_warm"])
    sbw = first["_warm"].get("sync_breakdown") or {}
    sbc = first.get("sync_breakdown") or {}
    res["warm_pool_pass"] = {k: warm.get(k) for k in ("tokens_per_s", "seconds", "handles_created", "create_ms", "map_calls", "sync_map_ms", "mapper_thread_map_ms",
                                                      "peak_concurrent_sequences", "roofline_prefill", "roofline_decode")}
    res["cold_over_warm_tokens_per_s"] = round(res["tokens_per_s"] / warm["tokens_per_s"], 4)
    # handle creation of the cold pass: all of it on the mapper thread except what a synchronous batch had to create itself
    res["handle_creation"] = {"handles": res.get("handles_created"), "total_ms": res.get("create_ms"), "on_critical_path_ms": sbc.get("create_ms"),
                              "warm_pass_total_ms": warm.get("create_ms"), "warm_pass_on_critical_path_ms": sbw.get("create_ms")}
    if with_cpu:
        pf, dc, L, Hq, Hkv, D = first["_stats"]
        res["cpu_baseline"] = cpu_baseline(dtype, L, Hq, Hkv, D, 16384, 1, first["tokens"], pf, dc, what.split(":")[0])
    if ab_deferred:
        second = digest(run(False))
        res["deferred_reclamation_off"] = {k: second[k] for k in ("tokens_per_s", "seconds", "peak_concurrent_sequences", "kv_util_time_weighted",
                                                                  "kv_live_over_mapped_mean_per_iteration", "map_calls", "unmap_calls",
                                                                  "sync_map_ms", "sync_fence_wait_ms", "mapper_thread_map_ms")}
    return res


def scale_series_leg(make_runner, mem_for_kv, tp, hook=None, max_over_ranks=None) -> dict:
    """ONE fixed workload for every N: Yi-34B (56 query / 8 kv heads, 60 layers) tensor-parallel over N GPUs (56/N
    and 8/N heads per rank), 2 MiB pages, static trace @ 131 072 ctx, P:D = 500, Sarathi 16 k chunks, ONE request end to end — the
    workload of configs[3] / run_figure_6.sh:32-33 at every degree, so that tokens/s over the N = 1 / 2 / 4 / 8 lines IS a strong-scaling
    series (the lines' `value`s are BASELINE.json's per-N configs: three different models).  N = 2 / 4 run exactly this as their main
    workload; N = 1 and N = 8 run it here, outside the timed region.  hook: the per-iteration control-plane exchange of an N > 1 job;
    max_over_ranks(dt): the job's time."""
    import torch
    from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers
    r = make_runner("yi-34b", tp, 131072, 2 << 20, 4, "fa_vattn", mem_for_kv)
    try:
        r.iter_hook = hook
        r.run_static_trace(1, 16384, 500.0, 16384)                   # warm-up: one 16 k request
        torch.cuda.synchronize()
        r.stats.__init__()
        enable_op_timers(True, every=1)
        t0 = time.perf_counter()
        r.run_static_trace(1, 131072, 500.0, 16384)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        det = drain_op_timers_detail()
        enable_op_timers(False)
        if max_over_ranks is not None:
            dt = max_over_ranks(dt)
        roofs = rooflines(det)
        tk = r.stats.prefill_tokens + r.stats.decode_tokens
        return {"workload": "yi-34b TP=%d (%d/%d heads per rank, 60 layers), 2 MiB pages, static trace @ 131072 ctx, P:D=500, Sarathi 16 k chunks, ONE request "
                            "(130810 prefill + 262 decode tokens): the same request at every N" % (tp, r.Hq, r.Hkv),
                "n_gpus": tp, "tokens": tk, "seconds": round(dt, 3), "tokens_per_s": round(tk / dt, 1),
                "prefill_frac": (roofs.get("roofline_prefill") or {}).get("frac"), "decode_frac": (roofs.get("roofline_decode") or {}).get("frac")}
    finally:
        r.close()


def c4_rank_share_leg(make_runner, mem_for_kv) -> dict:
    """The 128 k half of BASELINE.json's metric on ONE GPU: one rank's share of configs[3] (Yi-34B TP = 2: 28 query / 4 kv heads, 60
    layers, 2 MiB pages), static trace @ 131 072 ctx, P:D = 500, Sarathi 16 k chunks (scripts/benchmark_e2e_static_trace.py:6-57,
    artifact_asplos25/run_figure_6.sh:32-33), ONE request end to end: 8 prefill chunks against a growing prefix, then 262 decode steps
    of the single sequence.  The same code path as `bench.py --gpus 2` minus the control-plane exchange; a tensor-parallel job's
    tokens/s equals one rank's (every rank processes the same tokens with its head shard)."""
    import torch
    from vattention_amd import vattention
    from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers
    r = make_runner("yi-34b", 2, 131072, 2 << 20, 4, "fa_vattn", mem_for_kv)
    try:
        r.run_static_trace(1, 16384, 500.0, 16384)                   # warm-up: one 16 k request (kernels, plans, first handles)
        torch.cuda.synchronize()
        r.stats.__init__()
        enable_op_timers(True, every=1)
        t0 = time.perf_counter()
        r.run_static_trace(1, 131072, 500.0, 16384)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        det = drain_op_timers_detail()
        enable_op_timers(False)
        roofs = rooflines(det)
        tk = r.stats.prefill_tokens + r.stats.decode_tokens
        u = r.stats.kv_util_samples
        return {"workload": "one TP=2 rank's share of configs[3]: yi-34b (28/4 heads per rank, 60 layers), 2 MiB pages, static trace @ 131072 ctx, P:D=500, "
                            "Sarathi 16 k chunks, ONE request (130810 prefill + 262 decode tokens)",
                "tokens": tk, "seconds": round(dt, 3), "tokens_per_s": round(tk / dt, 1),
                "prefill_frac": (roofs.get("roofline_prefill") or {}).get("frac"), "decode_frac": (roofs.get("roofline_decode") or {}).get("frac"),
                "roofline_prefill": roofs.get("roofline_prefill"), "roofline_decode": roofs.get("roofline_decode"),
                "kv_live_over_mapped_mean": round(sum(u) / max(1, len(u)), 4), "kv_live_over_mapped_min": round(min(u), 4) if u else None,
                "external_fragmentation": r.stats.ext_frag_max}
    finally:
        r.close()


def hybrid_sarathi_leg(make_runner, mem_for_kv, chunk=4096) -> dict:
    """SURVEY §8 f1 on the reference's own scheduler shape (vattention_flashattention_pod_wrapper.py:121-203, pod_attn/tests/
    attn_sweep.py:82-97): Yi-6B, 64 requests of 8 192 tokens (P:D = 15: 7 680 prefill + 512 decode), Sarathi 4 k chunks — every prefill
    chunk rides with the decode batch of the sequences already running (up to 63).  The SAME trace twice: backend fa_vattn (prefill
    launch, then decode launch) and fa_streams (this package's hybrid policy: decode on a side stream beside an unsplit prefill when
    a host-side estimate says so).  A fused prefill || decode launch does not exist in the product (three measured designs lose to
    the serial order)."""
    import torch
    out = {"workload": "yi-6b TP=1, 64 requests x 8192 tokens (7680 prefill + 512 decode), Sarathi %d-token chunks with piggy-backed decodes, 32 layers" % chunk}
    for name, backend in (("serial", "fa_vattn"), ("streams", "fa_streams")):
        r = make_runner("yi-6b", 1, 8192, 2 << 20, 64, backend, min(mem_for_kv, 40 << 30))
        try:
            r.run_static_trace(2, 8192, 15.0, chunk)                     # warm-up
            torch.cuda.synchronize()
            r.stats.__init__()
            hyb = {"n": 0, "ov": 0}
            wr = r.wrapper
            if hasattr(wr, "_plan_overlap"):
                orig = wr._plan_overlap

                def counted(orig=orig):
                    v = orig()
                    hyb["n"] += bool(wr.prefill_query_lens and wr.decode_batch_size)
                    hyb["ov"] += bool(v)
                    return v
                wr._plan_overlap = counted
            t0 = time.perf_counter()
            r.run_static_trace(64, 8192, 15.0, chunk)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tk = r.stats.prefill_tokens + r.stats.decode_tokens
            out["tokens_per_s_" + name] = round(tk / dt, 1)
            out["seconds_" + name] = round(dt, 3)
            out["iterations_" + name] = r.stats.iterations
            if name == "streams":
                out["hybrid_iterations"], out["overlapped_iterations"] = hyb["n"], hyb["ov"]
                wr._plan_overlap = orig
        finally:
            r.close()
    out["streams_over_serial"] = round(out["tokens_per_s_streams"] / out["tokens_per_s_serial"], 4)
    return out


def open_loop_leg(make_runner, mem_for_kv, lengths, qps, requests) -> dict:
    """The reference's OPEN-loop arrival process (poisson_request_interval_generator.py:9-21, seed 42, interval capped at 3/qps) on one
    TP = 8 rank of Llama-3-70B (configs[4]: qps = 6), on a virtual clock: every iteration advances it by the measured GPU time of
    its attention + KV work plus a stated stand-in for the transformer body this harness does not run — per token 2 x 70e9 / 8
    flop at 1.0 PFLOP/s = 17.5 us, and never less than streaming the rank's weights once (70e9 x 2 B / 8 at 6 TB/s = 2.9 ms)."""
    r = make_runner("llama-3-70b", 8, 32768, 8 << 20, 256, "fa_vattn_megacache", mem_for_kv)
    try:
        out = r.run_dynamic_trace(requests, lengths=lengths, qps=qps, body_time=(17.5e-6, 2.9e-3))
    finally:
        r.close()
    ol = out["open_loop"]
    ol.update({"workload": "one TP=8 rank of llama-3-70b (8/1 heads, 80 layers), %d arxiv-length requests, Poisson arrivals at qps=%g (reference recipe, "
                           "seed 42), vLLM scheduler, megacache 8 MiB pages, admission look-ahead on" % (requests, qps),
               "peak_concurrent_sequences": out["peak_running"], "iterations": out["iters"], "tokens": out["tokens"],
               "gpu_seconds_wall": round(out["seconds"], 2), "tokens_per_virtual_s": round(out["tokens"] / ol["virtual_seconds"], 1),
               "sync_map_ms_total": round(out["sync_map_ms"], 1), "mapper_thread_map_ms": round(out["async_map_ms"], 1),
               "kv_util_time_weighted": out.get("kv_util_time_weighted")})
    return ol


def capacity_leg(dev, mem_for_kv) -> dict:
    """Whole-HBM KV capacity (BASELINE configs[4] 'KV-capacity stress'): one TP = 8 rank of Llama-3-70B (80 layers x 1 kv head), 256
    slots x 32 k, megacache with 8 MiB pages; every slot grows in steps until step() raises the reference's OOM error
    (vattention.cu:295).  Reports what was mapped against the budget, how long the fill took and how many handles back it."""
    import torch
    from vattention_amd import vattention
    L, kvh, D, B, ctx, page = 80, 1, 128, 256, 32768, 8 << 20
    vattention.enable_layered_async(False)
    ts = vattention.init_kvcache(L, kvh, D, B, ctx, dev.index or 0, torch.float16, page, True)
    try:
        npages = vattention.reserve_physical_pages(mem_for_kv)
        lay = vattention.layout()
        tpp = lay["tokens_per_page"]
        lens = [0] * B
        t0 = time.perf_counter()
        err, steps = None, 0
        grow = 8 * tpp
        while err is None:
            for i in range(B):
                lens[i] = min(ctx, lens[i] + grow)
            try:
                vattention.step(lens, False)
                steps += 1
            except RuntimeError as e:
                err = str(e)
            if all(x == ctx for x in lens) and err is None:
                break
        fill_s = time.perf_counter() - t0
        st, vs = vattention.state(), vattention.stats()
        mapped_groups = sum(st["mapped"])
        mapped_bytes = mapped_groups * 2 * page
        # first and last mapped token rows are readable and writable
        k = ts[0]
        last = max(0, min(ctx, st["mapped"][0] * tpp) - 1)
        k[0, 0].fill_(1.0)
        k[0, last].fill_(2.0)
        torch.cuda.synchronize()
        ok = bool(float(k[0, 0, 0, 0, 0]) == 1.0 and float(k[0, last, L - 1, 0, D - 1]) == 2.0)
        total_b = torch.cuda.mem_get_info(dev)[1]
        t1 = time.perf_counter()
    finally:
        vattention.cleanup()
    return {"workload": "one TP=8 rank of llama-3-70b: 80 layers x 1 kv head, 256 slots x 32768 tokens (320 GiB of virtual tensors), megacache 8 MiB pages "
                        "(%d tokens per page), all slots grown in steps of 8 pages until the allocator's OOM error" % tpp,
            "budget_bytes": int(mem_for_kv), "pool_pages": int(npages), "mapped_bytes": int(mapped_bytes),
            "mapped_over_budget": round(mapped_bytes / mem_for_kv, 4), "mapped_over_hbm": round(mapped_bytes / total_b, 4),
            "tokens_resident": int(mapped_groups * tpp), "fill_seconds": round(fill_s, 2), "grow_steps": steps,
            "handles_created": int(vs["handles_created"]), "create_ms": round(vs["create_ns"] / 1e6, 1), "map_calls": int(vs["map_calls"]),
            "oom_error": err, "first_and_last_rows_read_write": ok, "cleanup_seconds": round(time.perf_counter() - t1, 2)}


if __name__ == "__main__":
    main()
