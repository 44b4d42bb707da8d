#!/usr/bin/env python3
"""One-prompt and batched prefill launches of a TP8 rank (8 query / 1 kv head) through whatever work list the loaded library's planner
builds (one workgroup per piece), timed alone on resident tensors.  Used by tools/r05/r05_planner_ab.sh to A/B two planners."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd import flash_attn as FA  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv, D = 8, 1, 128
for q_lens in ([4119], [5000], [7341], [8192], [9441], [12000], [14000], [16384], [20751], [24000], [29092], [6526, 14505, 5364], [18684, 6354], [10159, 5174]):
    torch.manual_seed(1)
    P, ctx = len(q_lens), max(q_lens) + 64
    kc = torch.randn(P, ctx, Hkv, D, device=DEV, dtype=torch.float16)
    vc = torch.randn(P, ctx, Hkv, D, device=DEV, dtype=torch.float16)
    T = sum(q_lens)
    q = torch.randn(T, Hq, D, device=DEV, dtype=torch.float16)
    out = torch.empty_like(q)
    starts = torch.tensor([sum(q_lens[:i]) for i in range(P)], dtype=torch.int32, device=DEV)
    ql = torch.tensor(q_lens, dtype=torch.int32, device=DEV)
    idx = torch.arange(P, dtype=torch.int32, device=DEV)
    p = K.AttnParams()
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = P, max(q_lens), Hq, Hkv, D, 1
    pl = FA.prefill_plan(p, q_lens if P > 1 else None, q_lens, DEV, persistent=False)
    if P == 1:
        f = lambda: FA.flash_attn_with_kvcache(q.unsqueeze(0), kc, vc, cache_seqlens=ql, causal=True, out=out.unsqueeze(0), _max_seqlen_k=q_lens[0],
                                               _pf_plan=pl if pl.t is not None else "none")
    else:
        f = lambda: FA.flash_attn_varlen_with_kvcache(q, kc, vc, starts, ql, max(q_lens), ql, idx, causal=True, out=out, _max_seqlen_k=max(q_lens),
                                                      _pf_plan=pl if pl.t is not None else None)
    best = 1e9
    for rep in range(3):
        f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    fl = sum(4.0 * Hq * D * n * (n + 1) / 2 for n in q_lens)
    print("%-24s %.4f ms  %6.0f TFLOP/s  [%s]" % (q_lens, best, fl / best / 1e9, "list: %d pieces, %d split blocks" % (pl.n_items, pl.n_blocks) if pl.t is not None else "default launch"), flush=True)
