#!/usr/bin/env python3
"""Launch-bound decode iterations (small batch / short context): 32 layers of the decode call issued eagerly from Python vs
replayed as ONE captured HIP graph.  Per iteration (all layers), attention only.  usage: python tools/graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.flash_attn import flash_attn_with_kvcache  # noqa: E402

DEV = "cuda:0"
L = 32


def main():
    torch.zeros(1, device=DEV)
    for name, B, ctx, Hq, Hkv in [("B16 @ 2k", 16, 2048, 32, 8), ("B1 @ 8k", 1, 8192, 32, 8), ("B64 @ 1k", 64, 1024, 32, 8), ("B16 @ 32k", 16, 32768, 32, 4)]:
        kc = [torch.randn(B, ctx + 8, Hkv, 128, device=DEV).half() for _ in range(2)]      # two layers' worth of cache, reused
        vc = [torch.randn(B, ctx + 8, Hkv, 128, device=DEV).half() for _ in range(2)]
        q = torch.randn(B, 1, Hq, 128, device=DEV).half()
        kn = torch.randn(B, 1, Hkv, 128, device=DEV).half()
        vn = torch.randn(B, 1, Hkv, 128, device=DEV).half()
        cl = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
        out = torch.empty_like(q)

        def layers():
            for l in range(L):
                flash_attn_with_kvcache(q, kc[l & 1], vc[l & 1], kn, vn, cache_seqlens=cl, causal=True, out=out)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            layers()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            layers()

        def wall(fn, iters=20):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3

        with torch.cuda.stream(side):
            t_eager = wall(layers)
        t_graph = wall(g.replay)
        print("%-10s %d layers: eager %.3f ms   one graph replay %.3f ms   %.2fx" % (name, L, t_eager, t_graph, t_eager / t_graph))


if __name__ == "__main__":
    main()
