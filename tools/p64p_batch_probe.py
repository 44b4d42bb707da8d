#!/usr/bin/env python3
"""Where do the persistent prefill queues lose inside the replay?  (round 5)  The TP8 rank's batched prompts (8 query / 1 kv head, the
replay's first admission batches) as ONE varlen launch each, timed alone: contiguous K/V against one layer's view of a megacache tensor
(80 layers: rows 20 KiB apart, what the replay legs run), one workgroup per piece against host-assigned and drawn persistent queues.
usage: python tools/p64p_batch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd import flash_attn as FA  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv, D = 8, 1, 128


def case(q_lens, layers):
    torch.manual_seed(1)
    P, ctx = len(q_lens), max(q_lens) + 64
    if layers > 1:
        kc = torch.randn(P, ctx, layers, Hkv, D, device=DEV, dtype=torch.float16)[:, :, layers // 2]
        vc = torch.randn(P, ctx, layers, Hkv, D, device=DEV, dtype=torch.float16)[:, :, layers // 2]
    else:
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
    plans = {"per piece": FA.prefill_plan(p, q_lens, q_lens, DEV, persistent=False),
             "assigned queues": FA.prefill_plan(p, q_lens, q_lens, DEV, persistent=True, drawn=False),
             "drawn queues": FA.prefill_plan(p, q_lens, q_lens, DEV, persistent=True, drawn=True)}
    res = {k: [] for k in plans}
    for rep in range(3):
        for name, pl in plans.items():
            f = lambda: FA.flash_attn_varlen_with_kvcache(q, kc, vc, starts, ql, max(q_lens), ql, idx, causal=True, out=out, _max_seqlen_k=max(q_lens),
                                                          _pf_plan=pl if pl.t is not None else None)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
    fl = sum(4.0 * Hq * D * n * (n + 1) / 2 for n in q_lens)
    base = min(res["per piece"])
    print("%-34s %-12s" % (q_lens, "megacache x%d" % layers if layers > 1 else "contiguous") + "  ".join(
        "%s %.4f ms (%.0f TF, %.3f)" % (k, min(v), fl / min(v) / 1e9, base / min(v)) for k, v in res.items()) +
        "  [%d pieces, %d workgroups]" % (plans["assigned queues"].n_items, plans["assigned queues"].n_wg), flush=True)


if __name__ == "__main__":
    for q_lens in ([6526, 14505, 5364], [20751], [7344, 8347, 7339, 5353], [9441], [5602, 17010, 7224]):
        for layers in (1, 80):
            case(q_lens, layers)
