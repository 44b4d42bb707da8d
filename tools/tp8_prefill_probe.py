#!/usr/bin/env python3
"""Where does the prefill time of a tensor-parallel shard's batched prompts go?  Times, on one TP=8 rank's head shape (8 / 1 heads), the
batched variable-length launches of the first prefill iterations of the reference's dynamic trace (tests/golden/c3_arxiv_lengths_256.json,
vLLM scheduler: whole prompts packed into 32 768 tokens) and compares each with the tile-step model (W key-tile steps over 256 CUs at the
per-step time of a chip-filling launch) and with the same prompts launched one by one.  usage: python tools/tp8_prefill_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv, D = 8, 1, 128
if len(sys.argv) > 2:        # other head shapes: python tools/tp8_prefill_probe.py 32 8  (Llama-3-8B)
    Hq, Hkv = int(sys.argv[1]), int(sys.argv[2])


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.zeros(1, device=DEV)
    reqs = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
    pre = [p for p, _ in reqs]
    its, cur, bud = [], [], 32768
    for p in pre:
        if p > bud:
            its.append(cur)
            cur, bud = [], 32768
        cur.append(p)
        bud -= p
    # reference rate: one chip-filling launch (a 29 k prompt)
    n = 29092
    q = torch.randn(1, n, Hq, D, device=DEV, dtype=torch.float16)
    k = torch.randn(1, n, Hkv, D, device=DEV, dtype=torch.float16)
    v = torch.randn(1, n, Hkv, D, device=DEV, dtype=torch.float16)
    cl = torch.tensor([n], dtype=torch.int32, device=DEV)
    t_big = timeit(lambda: flash_attn_with_kvcache(q, k, v, cache_seqlens=cl, causal=True, _max_seqlen_k=n), 5)
    steps_big = Hq * sum(4 * (qb + 1) for qb in range((n + 255) // 256)) / 256.0
    us_per_step = t_big * 1e3 / steps_big
    print("29092-token prompt alone: %.3f ms = %.0f TFLOP/s; %.1f tile steps per CU -> %.2f us per step" % (
        t_big, 4.0 * Hq * D * n * (n + 1) / 2 / t_big / 1e9, steps_big, us_per_step))
    for lens in its[:8] + [[7344], [4119, 4500], [12001, 900, 600], [5963, 16991], [5637, 23774], [23774, 5637, 1000], [23774], [5637], [16991, 16991]]:
        T, B = sum(lens), len(lens)
        ctx = max(lens)
        q = torch.randn(T, Hq, D, device=DEV, dtype=torch.float16)
        kc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        vc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        out = torch.empty_like(q)
        starts = torch.tensor([sum(lens[:i]) for i in range(B)], dtype=torch.int32, device=DEV)
        ql = torch.tensor(lens, dtype=torch.int32, device=DEV)
        idx = torch.arange(B, dtype=torch.int32, device=DEV)
        if B > 1:
            t_batch = timeit(lambda: flash_attn_varlen_with_kvcache(q, kc, vc, starts, ql, max(lens), ql, idx, causal=True, out=out, _max_seqlen_k=max(lens)))
        else:
            cl1 = torch.tensor(lens, dtype=torch.int32, device=DEV)
            t_batch = timeit(lambda: flash_attn_with_kvcache(q.unsqueeze(0), kc, vc, cache_seqlens=cl1, causal=True, out=out.unsqueeze(0), _max_seqlen_k=lens[0]))
        t_list = float("nan")
        if B > 1:        # the same launch driven by the planner's work list (compact for ragged batches, cut where that pays)
            from vattention_amd import flash_attn as FA
            from vattention_amd import kernels as K
            pp = K.AttnParams()
            pp.b, pp.seqlen_q, pp.h, pp.h_k, pp.d, pp.is_causal = B, max(lens), Hq, Hkv, D, 1
            pl = FA.prefill_plan(pp, lens, lens, DEV)
            if pl.t is not None:
                t_list = timeit(lambda: flash_attn_varlen_with_kvcache(q, kc, vc, starts, ql, max(lens), ql, idx, causal=True, out=out, _max_seqlen_k=max(lens), _pf_plan=pl))
        t_single = 0.0
        tok = 0
        singles = []
        for i, nn in enumerate(lens):
            qi, oi = q[tok:tok + nn].unsqueeze(0), out[tok:tok + nn].unsqueeze(0)
            cli = torch.tensor([nn], dtype=torch.int32, device=DEV)
            singles.append(timeit(lambda: flash_attn_with_kvcache(qi, kc[i:i + 1], vc[i:i + 1], cache_seqlens=cli, causal=True, out=oi, _max_seqlen_k=nn)))
            t_single += singles[-1]
            tok += nn
        W = sum(Hq * sum(4 * (qb + 1) for qb in range((x + 255) // 256)) for x in lens)
        longest = max(4 * ((x + 255) // 256) for x in lens)
        model = max(W / 256.0, longest) * us_per_step / 1e3
        fl = sum(4.0 * Hq * D * x * (x + 1) / 2 for x in lens)
        print("%-34s one launch %.3f ms (%.0f TFLOP/s), with the work list %.3f ms | one by one %.3f ms %s | model %.3f ms (W/256 = %.0f steps, longest block %d) -> launch / model %.2f" % (
            lens, t_batch, fl / t_batch / 1e9, t_list, t_single, [round(x, 3) for x in singles], model, W / 256.0, longest, t_batch / model))


if __name__ == "__main__":
    main()
