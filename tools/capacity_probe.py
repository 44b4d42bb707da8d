#!/usr/bin/env python3
"""Capacity stress in the shape of BASELINE configs[4] (Llama-3-70B, TP8: one rank = 80 layers, 1 kv head, d 128): give the
allocator 0.9 x HBM minus the rank's weight shard, grow 256 slots until the pool is empty, report how much of the 288 GB is
mapped behind the virtual tensors, what mapping cost, and that kernels can touch the first and last mapped rows.
usage: python tools/capacity_probe.py [--page-kib 2048] [--megacache] [--weights-gib 17.5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd import vattention  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--page-kib", type=int, default=2048)
ap.add_argument("--megacache", action="store_true")
ap.add_argument("--weights-gib", type=float, default=17.5)
ap.add_argument("--layers", type=int, default=80)
ap.add_argument("--kv-heads", type=int, default=1)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--ctx", type=int, default=32768)
a = ap.parse_args()

torch.zeros(1, device="cuda:0")
free_b, total_b = torch.cuda.mem_get_info(0)
page = a.page_kib << 10
t0 = time.perf_counter()
tensors = vattention.init_kvcache(a.layers, a.kv_heads, 128, a.batch, a.ctx, 0, torch.float16, page, a.megacache)
t_init = time.perf_counter() - t0
budget = min(int(total_b * 0.9), free_b) - int(a.weights_gib * (1 << 30))
t0 = time.perf_counter()
npages = vattention.reserve_physical_pages(budget)
t_reserve = time.perf_counter() - t0
lay = vattention.layout()
tpp = lay["tokens_per_page"]
pages_per_group = 2 if a.megacache else 2 * a.layers
groups = npages // pages_per_group
max_groups_per_seq = (a.ctx + tpp - 1) // tpp
lens = [0] * a.batch
grown = 0
t0 = time.perf_counter()
oom = None
step_ms = []
while True:
    progressed = False
    for r in range(a.batch):
        if lens[r] + tpp <= a.ctx:
            trial = list(lens)
            trial[r] += tpp
            ts = time.perf_counter()
            try:
                vattention.step(trial, False)
            except RuntimeError as e:
                oom = str(e)
                break
            step_ms.append((time.perf_counter() - ts) * 1e3)
            lens = trial
            grown += 1
            progressed = True
    if oom or not progressed:
        break
t_map = time.perf_counter() - t0
vattention.wait()
c = vattention.counts()
st = vattention.stats()
free_after = vattention.num_free_kvblocks()
mapped_bytes = c["mapped_groups"] * pages_per_group * page
# touch the first and the last mapped token row of the first and last layer's K and V
ok = True
probe = [(0, 0)] + [(r, lens[r] - 1) for r in range(a.batch) if lens[r] > 0][-1:]
for ti in (0, len(tensors) // 2 - 1, len(tensors) // 2, len(tensors) - 1):
    for r, tok in probe:
        row = tensors[ti][r, tok]
        row.fill_(1.5)
        torch.cuda.synchronize()
        ok = ok and bool((row == 1.5).all().item())
hbm_free_after, _ = torch.cuda.mem_get_info(0)
out = {
    "config": {"layers": a.layers, "kv_heads": a.kv_heads, "head_dim": 128, "max_batch_size": a.batch, "max_ctx": a.ctx,
               "page_kib": a.page_kib, "megacache": a.megacache, "tokens_per_page": tpp},
    "hbm_total_gb": total_b / 1e9, "kv_budget_gb": budget / 1e9, "pool_pages": npages, "pool_groups": groups,
    "virtual_reserved_gib": len(tensors) * tensors[0].stride(0) * tensors[0].element_size() * a.batch / (1 << 30),
    "grow_steps": grown, "oom_message": oom, "num_free_kvblocks_at_end": free_after,
    "mapped_groups": c["mapped_groups"], "mapped_gb": mapped_bytes / 1e9, "mapped_over_hbm": mapped_bytes / total_b,
    "mapped_over_budget": mapped_bytes / budget, "tokens_resident": sum(lens), "slots_used": sum(1 for x in lens if x),
    "hbm_free_after_gb": hbm_free_after / 1e9,
    "seconds": {"init_kvcache": t_init, "reserve_physical_pages": t_reserve, "grow_until_oom": t_map},
    "map_calls": st["map_calls"], "per_group_ms_first_100": sum(step_ms[:100]) / max(1, len(step_ms[:100])),
    "per_group_ms_last_100": sum(step_ms[-100:]) / max(1, len(step_ms[-100:])),
    "first_and_last_rows_readable_writable": ok,
}
t0 = time.perf_counter()
vattention.cleanup()
out["seconds"]["cleanup"] = time.perf_counter() - t0
print(json.dumps(out))
