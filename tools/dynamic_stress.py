#!/usr/bin/env python3
"""KV-capacity / fragmentation stress (BASELINE.json configs[2] / [4] shape): many concurrent sequences, small pages,
async mapping, pool sized so that admission control, on-demand reclamation and the TLB-invalidation path all run.
usage: python tools/dynamic_stress.py [--model llama-3-8b] [--page-kib 64] [--batch 256] [--requests 256] [--pool-gib N] [--layers L]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3-8b")
ap.add_argument("--page-kib", type=int, default=64)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--requests", type=int, default=256)
ap.add_argument("--pool-gib", type=float, default=0.0, help="0 = 0.9 x HBM minus 12 GiB")
ap.add_argument("--layers", type=int, default=0, help="simulate fewer layers (allocator stress stays identical per layer)")
ap.add_argument("--tp", type=int, default=1)
ap.add_argument("--passes", type=int, default=1, help="replay the trace this many times on the same allocator (pass 2+ = warm handle pool)")
ap.add_argument("--megacache", action="store_true", help="one page covers all layers (2 handles per page-group instead of 2L)")
a = ap.parse_args()
_fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c3_arxiv_lengths_256.json")
LENGTHS = json.load(open(_fx))["requests"] if os.path.exists(_fx) and not os.environ.get("VATTN_FITTED_LENGTHS") else None
torch.zeros(1, device="cuda")
model = ModelConfig.named(a.model, dtype=torch.float16, max_model_len=32768, attention_backend="fa_vattn_megacache" if "--megacache" in sys.argv else "fa_vattn")
if a.layers:
    model.num_layers = a.layers
free_b, total_b = torch.cuda.mem_get_info()
pool = int(a.pool_gib * (1 << 30)) if a.pool_gib else min(int(total_b * 0.9), free_b) - (12 << 30)
r = HotPathRunner(model, ParallelConfig(a.tp, 1), CacheConfig(page_size=a.page_kib << 10, max_batch_size=a.batch, memory_for_gpu=pool))
try:
    for ps in range(a.passes):
        r.stats.__init__()
        out = r.run_dynamic_trace(a.requests, lengths=LENGTHS)
        out["lengths_from"] = "tests/golden/c3_arxiv_lengths_256.json (reference recipe)" if LENGTHS else "fitted log-normals"
        out["pass"] = ps + 1
        if ps + 1 < a.passes:
            print(json.dumps(out), flush=True)
            import vattention_amd.vattention as _v
            _v.step([0] * a.batch, True)          # eager reclaim: unmap everything, handles stay created (warm pool)
    out.update({"model": a.model, "tp": a.tp, "layers": r.L, "page_kib": a.page_kib, "megacache": a.megacache, "max_batch_size": a.batch, "pool_gib": round(pool / (1 << 30), 1)})
    print(json.dumps(out))
finally:
    r.close()
