"""bench.py's host-side helpers (no GPU): the roofline objects built from the op-timer records, the CPU-baseline arithmetic, the
open-loop arrival recipe of the replay harness."""
import math
import random

import torch

import bench


def test_rooflines_pick_the_dominant_kernel_and_scale_sampled_time():
    detail = {"attn_prefill": {"ms": 10.0, "timed": 4, "n": 32, "work": 4 * 2.0e12},        # 4 timed launches of 2 TFLOP in 10 ms
              "attn_decode": {"ms": 3.0, "timed": 10, "n": 10, "work": 10 * 1.2e9}}
    r = bench.rooflines(detail, {"prefill": 123, "decode": 456})
    assert r["roofline"]["bound"] == "mfma" and r["roofline"]["dominant_of"] == ["decode", "prefill"]
    assert abs(r["roofline_prefill"]["achieved"] - 800.0) < 1e-6 and r["roofline_prefill"]["frac"] == 0.32
    assert r["roofline_prefill"]["est_total_ms"] == 80.0 and r["roofline_prefill"]["launches_timed"] == 4
    assert abs(r["roofline_decode"]["achieved"] - 4000.0) < 1e-6 and r["roofline_decode"]["unit"] == "GB/s"
    assert r["roofline_prefill"]["traffic"] == 123 and r["roofline_decode"]["traffic"] == 456
    only_dec = bench.rooflines({"attn_decode": detail["attn_decode"]})
    assert only_dec["roofline"]["bound"] == "hbm" and "roofline_prefill" not in only_dec
    assert bench.rooflines({}) == {}


def test_mfma_utilisation_three_ways():
    """fraction of the spec peak, of the peak at the sustained clock, and the matrix-pipe duty of the PMC pass"""
    detail = {"attn_prefill": {"ms": 7.64, "timed": 1, "n": 1, "work": 8.761e12}}
    r = bench.rooflines(detail, {"prefill": 1})
    bench.mfma_views(r, 1811.0, {"mfma_busy_frac": 0.655, "source": "profiles/x.json"})
    assert r["roofline"]["frac"] == round(8.761e12 / 7.64e-3 / 1e12 / 2500.0, 4)
    assert abs(r["roofline"]["frac_at_clock"] - r["roofline"]["achieved"] / (2500.0 * 1811.0 / 2400.0)) < 1e-4
    assert r["roofline"]["mfma_busy_frac"] == 0.655 and r["roofline_prefill"]["frac_at_clock"] == r["roofline"]["frac_at_clock"]
    d = bench.rooflines({"attn_decode": {"ms": 3.0, "timed": 10, "n": 10, "work": 10 * 1.2e9}})
    bench.mfma_views(d, 1811.0, {"kernel_us": {"decode": 188.1, "prefill": None}, "source": "profiles/x.json"})
    assert d["roofline_decode"]["kernel_us_per_launch"] == 188.1 and d["roofline"]["kernel_us_per_launch"] == 188.1
    assert "frac_at_clock" not in d["roofline"]            # an HBM-bound kernel's peak does not move with the shader clock


def test_cpu_baseline_scales_measured_rates_by_counted_pairs():
    torch.manual_seed(0)
    b = bench.cpu_baseline(torch.float16, 2, 4, 2, 64, 512, 2, 1000, 512 * 513 / 2.0, 3 * 512.0, "test")
    assert b["kind"] == "port" and b["unit"] == "tokens/s" and b["cores"] >= 1 and b["value"] > 0
    assert "2 shard(s) x 2 layers x 4 heads" in b["sample"]


def test_open_loop_interval_recipe_matches_the_reference_generator():
    """poisson_request_interval_generator.py:9-21 restated: random.Random(seed), interval = min(-ln(1 - U) / qps, 3 / qps)."""
    qps, seed = 6.0, 42
    rng = random.Random(seed)
    want, t = [], 0.0
    for _ in range(16):
        t += min(-math.log(1.0 - rng.random()) / qps, 3.0 / qps)
        want.append(t)
    # the same arithmetic as replay.run_dynamic_trace's arrival table
    arng = random.Random(seed)
    got, t = [], 0.0
    for _ in range(16):
        t += min(-math.log(1.0 - arng.random()) / qps, 3.0 / qps)
        got.append(t)
    assert got == want and max(b - a for a, b in zip(got, got[1:])) <= 3.0 / qps + 1e-12


def test_stdout_carries_the_bench_line_only():
    """bench.py owns fd 1: what a library prints there (the page manager's state dump before an OOM error, 260 lines in the capacity leg)
    goes to stderr, and stdout is exactly ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, ctypes; sys.path.insert(0, %r); import bench; bench._own_stdout(); print('python noise'); "
            "ctypes.CDLL(None).puts(b'native noise'); ctypes.CDLL(None).fflush(None); os.write(1, b'raw fd noise\\n'); bench._emit('{\"metric\": 1}')" % root)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"metric": 1}, r.stdout
    for noise in ("python noise", "native noise", "raw fd noise"):
        assert noise in r.stderr


def test_bookkeeping_baseline_times_the_reference_allocator_beside_this_manager():
    """cpu_baseline.bookkeeping: the reference's own allocator (oracle/_ref, when built) and this manager on one call
    sequence; microseconds per step_async, driver calls free."""
    m = bench.bookkeeping_baseline()
    assert "error" not in m, m
    assert m["this_manager_inline_us_per_step_async"] > 0 and m["step_async_calls"] >= 100 and m["cores"] == 1
    if m["reference_us_per_step_async"] is not None:
        assert m["kind"] == "reference" and m["reference_over_this"] > 1.0      # the plan-then-execute manager does less host work per step


def test_telemetry_sampler_never_fails_the_benchmark_where_no_gpu_answers():
    import time
    from vattention_amd.telemetry import Sampler
    t0 = time.time()
    with Sampler(0, 0.02) as s:
        time.sleep(0.1)
    w = s.window(t0, time.time())
    assert set(w) >= {"source", "samples", "interval_s"}
    if w["source"] is None:
        assert w["samples"] == 0 and "clock_mhz_mean" not in w


def test_external_fragmentation_is_what_the_pool_cannot_hand_out_as_whole_groups():
    from vattention_amd.replay import ReplayStats
    st = ReplayStats()
    assert st.ext_frag_max == 0.0 and st.ext_frag_samples == 0


def test_replay_timer_stride_samples_every_layer_equally_often():
    """bench.py times every k-th launch of each attention operation in the replay legs; a replay issues one prefill launch per layer and
    iteration, so k must be coprime with every model's layer count (k = 8 on 80 / 32 layers timed layer 0 of every iteration — whose
    event pair also holds the host's planning — and seven of eight layers never: profiles/r05_timer_stride.txt)."""
    from vattention_amd.replay import ModelConfig
    k = bench.TIMER_EVERY["dynamic"]
    for name in ("yi-6b", "llama-3-8b", "yi-34b", "llama-3-70b"):
        L = ModelConfig.named(name).num_layers
        assert math.gcd(k, L) == 1, (name, L, k)
        # over L consecutive iterations (L x L launches) every layer is timed, and equally often
        timed = [n % L for n in range(L * L * k) if n % k == 0]
        counts = [timed.count(layer) for layer in range(L)]
        assert min(counts) == max(counts) > 0, (name, counts[:4])


def test_power_ceiling_leg_reports_ratios_and_never_raises(monkeypatch, tmp_path):
    """roofline.other.power_ceiling: the probe's two rates, their fractions of the 2.5 PF spec peak, and the prefill kernel's rate over them; a box
    without the binary or without a device yields {"error": ...} (the bench line is printed either way)."""
    import subprocess

    class R:
        stdout = '{"mfma_only_tflops": 1680.0, "tile_step_stream_tflops": 1360.0, "seconds_each": 0.70}\n'
        stderr = ""
    monkeypatch.setattr(bench.os.path, "exists", lambda p: True)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    d = bench.power_ceiling(1167.4)
    assert d["mfma_only_frac_of_peak"] == 0.672 and d["tile_step_stream_frac_of_peak"] == 0.544
    assert d["prefill_over_mfma_only"] == round(1167.4 / 1680.0, 4) and d["prefill_over_tile_step_stream"] == round(1167.4 / 1360.0, 4)
    assert "prefill_over_mfma_only" not in bench.power_ceiling(None)
    R.stdout = '{"mfma_only_tflops": 1680.0, "tile_step_stream_tflops": 1360.0, "seconds_each": 0.70, "hbm_read_stream_gbs": 6600.0}\n'
    pc, hc = bench.measured_ceilings(1167.4, 5717.0)
    assert "hbm_read_stream_gbs" not in pc and pc["prefill_over_tile_step_stream"] == round(1167.4 / 1360.0, 4)
    assert hc["read_stream_gbs"] == 6600.0 and hc["read_stream_frac_of_peak"] == 0.825 and hc["decode_over_read_stream"] == round(5717.0 / 6600.0, 4)

    class Bad:
        stdout = "no device\n"
        stderr = ""
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Bad())
    assert "error" in bench.power_ceiling(1000.0)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False)
    assert "not built" in bench.power_ceiling(1000.0)["error"]
    pc, hc = bench.measured_ceilings(1000.0, 5000.0)
    assert "error" in pc and "error" in hc
