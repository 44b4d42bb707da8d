#!/usr/bin/env python3
"""End-to-end effect of the two-stream hybrid backend on a Sarathi-scheduled static trace (chunked prefill with the running
decodes piggy-backed on every chunk): tokens/s of the attention + KV hot path under fa_vattn (serial) and fa_streams.
usage: python tools/hybrid_e2e.py [--model llama-3-8b] [--ctx 16384] [--chunk 1024] [--batch 32] [--pd 20] [--layers 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig  # noqa: E402


def run(a, backend):
    model = ModelConfig.named(a.model, dtype=torch.float16, max_model_len=a.ctx, attention_backend=backend)
    model.num_layers = a.layers
    cache = CacheConfig(page_size=2 << 20, max_batch_size=a.batch, memory_for_gpu=64 << 30)
    r = HotPathRunner(model, ParallelConfig(1, 1), cache, device="cuda:0")
    r.sample_kv_util = False
    try:
        r.run_static_trace(a.batch, a.ctx, a.pd, a.chunk)        # warm-up: maps the pages, builds the handle pool
        torch.cuda.synchronize()
        r.stats.__init__()
        t0 = time.perf_counter()
        r.run_static_trace(a.batch, a.ctx, a.pd, a.chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tok = r.stats.prefill_tokens + r.stats.decode_tokens
        return tok / dt, dt, r.stats.iterations
    finally:
        r.close()


if __name__ == "__main__":
    from vattention_amd.attention.vattention_flashattention_pod_wrapper import VAttentionFlashAttentionPodWrapper as _Pod
    _Pod.FUSED_ENABLED = True      # fa_pod = the LAB library's fused launch here (the product has none: see the wrapper)
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--ctx", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pd", type=float, default=20.0)
    ap.add_argument("--layers", type=int, default=8)
    a = ap.parse_args()
    print("model %s ctx %d chunk %d batch %d P:D %g layers %d" % (a.model, a.ctx, a.chunk, a.batch, a.pd, a.layers))
    res = {}
    for backend in ("fa_vattn", "fa_streams", "fa_pod", "fa_vattn", "fa_streams", "fa_pod"):
        tps, dt, it = run(a, backend)
        res.setdefault(backend, []).append(tps)
        print("  %-11s %9.0f tokens/s  (%.2f s, %d iterations)" % (backend, tps, dt, it))
    print("  two streams / serial = %.3f   fused launch (fa_pod) / serial = %.3f" % (max(res["fa_streams"]) / max(res["fa_vattn"]), max(res["fa_pod"]) / max(res["fa_vattn"])))
