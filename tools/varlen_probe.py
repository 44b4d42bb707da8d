#!/usr/bin/env python3
"""Several short prompts in one scheduler iteration: one attention launch per prompt (what the reference's wrapper does) vs
ONE batched variable-length launch.  Per layer, attention only (the per-prompt cache_flat launches are the same in both).
usage: python tools/varlen_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
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
    torch.manual_seed(0)
    for name, Hq, Hkv, lens in [("llama8b 8 x 512", 32, 8, [512] * 8), ("llama8b 16 x 256", 32, 8, [256] * 16),
                                ("llama8b ragged 100..1500", 32, 8, [100, 1500, 320, 777, 64, 1024, 250, 512]),
                                ("yi6b 4 x 2048", 32, 4, [2048] * 4), ("llama70b/tp8 8 x 512", 8, 1, [512] * 8)]:
        B, T, D, ctx = len(lens), sum(lens), 128, 2048
        q = torch.randn(T, Hq, D, device=DEV, dtype=torch.float16)
        kc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        vc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        out = torch.empty_like(q)
        starts = [sum(lens[:i]) for i in range(B)]
        i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
        st, ql, cl, idx = i32(starts), i32(lens), i32(lens), i32(list(range(B)))
        cls = [cl[i:i + 1] for i in range(B)]

        def per_prompt():
            for i, (s0, n) in enumerate(zip(starts, lens)):
                flash_attn_with_kvcache(q[s0:s0 + n].view(1, n, Hq, D), kc[i].unsqueeze(0), vc[i].unsqueeze(0), cache_seqlens=cls[i],
                                        causal=True, out=out[s0:s0 + n].view(1, n, Hq, D), _max_seqlen_k=n)

        def batched():
            flash_attn_varlen_with_kvcache(q, kc, vc, st, ql, max(lens), cl, idx, causal=True, out=out, _max_seqlen_k=max(lens))

        a, b = timeit(per_prompt), timeit(batched)
        fl = sum(4.0 * Hq * D * n * (n + 1) / 2 for n in lens)
        print("%-28s per-prompt launches %.3f ms (%5.0f TFLOP/s)   one batched launch %.3f ms (%5.0f TFLOP/s)   %.2fx" % (
            name, a, fl / a / 1e9, b, fl / b / 1e9, a / b))


if __name__ == "__main__":
    main()
