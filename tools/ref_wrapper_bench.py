#!/usr/bin/env python3
"""Does the PLANNED performance reach a caller that only has the reference's API?  

Drives two of bench.py's workloads twice on one GPU — once through this package's wrapper + cache engine (which pass the kernels
host-side hints: lengths, per-iteration plans, relaunch blocks), once through the REFERENCE's own, unmodified
VAttentionFlashAttentionWrapper + vATTNCacheEngine classes (tests/ref_loader.py: /root/reference where present, else the byte-compiled
copies under oracle/_ref/pyref) over the drop-in `vattention`, `flash_attn`, `sarathi.cache_ops` modules — and prints tokens/s side by
side.  The reference's wrapper hands the kernels device tensors only (vattention_flashattention_wrapper.py:159-166,194-205): what
it gets is what the library plans ON ITS OWN (decode: the device-planned stream decomposition; prefill: the lengths of
vattention.step_async resolved from the cache pointer).

  static        BASELINE.json configs[1]: Yi-6B TP=1, 2 MiB pages, one max_batch_size = 16 wave (16 x 32 702-token prefill + 65 batch-16
                decode iterations, 32 layers)
  dynamic_tp8   one TP=8 rank of Llama-3-70B (8 / 1 heads, 80 layers), the 256 arxiv-length requests closed loop, decode capped at 768

usage: python tools/ref_wrapper_bench.py [static] [dynamic_tp8] [--layers N]   (writes one JSON object per workload)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import ref_loader  # noqa: E402
from vattention_amd import vattention  # noqa: E402
from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers  # noqa: E402
from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig  # noqa: E402

DEV = torch.device("cuda:0")


class _Ref:
    def __init__(self, ctx):
        self.wrapper, self.engine_cls = ctx.wrapper, ctx.engine_mod.vATTNCacheEngine


def run(which: str, impl: str, layers: int, mem: int, lengths):
    ref_ctx = ref_loader.loaded() if impl == "reference" else None
    ctx = ref_ctx.__enter__() if ref_ctx else None
    try:
        if which == "static":
            model = ModelConfig.named("yi-6b", dtype=torch.float16, max_model_len=32768, attention_backend="fa_vattn")
            cache = CacheConfig(page_size=2 << 20, max_batch_size=16, memory_for_gpu=mem, vattn_keep_layout=True)
            tp = 1
        else:
            model = ModelConfig.named("llama-3-70b", dtype=torch.float16, max_model_len=32768, attention_backend="fa_vattn_megacache")
            cache = CacheConfig(page_size=8 << 20, max_batch_size=256, memory_for_gpu=mem, vattn_keep_layout=True)
            tp = 8
        if layers:
            model.num_layers = layers
        r = HotPathRunner(model, ParallelConfig(tp, 1), cache, device=str(DEV), reference=_Ref(ctx) if ctx else None)
        try:
            def step():
                r.stats.__init__()
                if which == "static":
                    r.run_static_trace(16, 32768, 500.0, None)
                else:
                    r.run_dynamic_trace(256, lengths=lengths)
                return r.stats.prefill_tokens + r.stats.decode_tokens
            r.sample_kv_util = False
            step()                                       # warm-up (cold pool, plan caches)
            torch.cuda.synchronize()
            enable_op_timers(True, every=1 if which == "static" else 7)
            t0 = time.perf_counter()
            tokens = step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            det = drain_op_timers_detail()
            enable_op_timers(False)
            ops = {k: {"launches": v["n"], "timed": v["timed"], "ms_per_launch": round(v["ms"] / v["timed"], 4) if v["timed"] else None,
                       "est_total_ms": round(v["ms"] * v["n"] / v["timed"], 1) if v["timed"] else None} for k, v in det.items()}
            import vattention_amd.flash_attn as FA
            return {"impl": impl, "shim_counters": dict(FA.counters), "tokens": tokens, "seconds": round(dt, 3), "tokens_per_s": round(tokens / dt, 1), "ops": ops,
                    "wrapper_class": type(r.wrapper).__module__ + "." + type(r.wrapper).__name__,
                    "engine_class": type(r.engine).__module__ + "." + type(r.engine).__name__}
        finally:
            r.close()
    finally:
        if ref_ctx:
            ref_ctx.__exit__(None, None, None)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    layers = int(sys.argv[sys.argv.index("--layers") + 1]) if "--layers" in sys.argv else 0
    which = args or ["static", "dynamic_tp8"]
    torch.zeros(1, device=DEV)
    free_b, total_b = torch.cuda.mem_get_info(DEV)
    mem = min(int(total_b * 0.9), free_b) - (12 << 30)
    lengths = [[pre, min(dec, 768)] for pre, dec in json.load(open(os.path.join(ROOT, "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]]
    if not ref_loader.available():
        raise SystemExit("reference wrapper / engine not available (oracle/_ref/pyref missing: run __graft_entry__.build() where /root/reference exists)")
    for w in which:
        res = {}
        for impl in ("repo", "reference", "repo", "reference"):      # alternate: boxes drift by a percent or two
            out = run(w, impl, layers, mem, lengths)
            res.setdefault(impl, []).append(out)
            print("  [%s] %-9s %9.1f tokens/s  (%.3f s)" % (w, impl, out["tokens_per_s"], out["seconds"]), file=sys.stderr, flush=True)
        best = {k: max(v, key=lambda o: o["tokens_per_s"]) for k, v in res.items()}
        print(json.dumps({"workload": w, "layers_override": layers or None,
                          "repo_tokens_per_s": [o["tokens_per_s"] for o in res["repo"]], "reference_tokens_per_s": [o["tokens_per_s"] for o in res["reference"]],
                          "reference_over_repo": round(best["reference"]["tokens_per_s"] / best["repo"]["tokens_per_s"], 4),
                          "repo": best["repo"], "reference": best["reference"]}), flush=True)


if __name__ == "__main__":
    main()
