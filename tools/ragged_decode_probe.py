#!/usr/bin/env python3
"""Ragged decode batch on one TP=8 rank (8 / 1 heads, 256 sequences whose contexts are the prompt lengths of the reference's dynamic
trace + 100 tokens): the uniform split (every sequence the same number of splits) against the length-balanced plan (vattn_decode_plan) at
several piece lengths, and against an equal-length batch of the same total size.  usage: python tools/ragged_decode_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.flash_attn import flash_attn_with_kvcache  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv, D = 8, 1, 128
LEGACY = 1 << 19      # variant bit: the grid heuristics / host plans of rounds 1-3


def timeit(fn, iters=30):
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
    torch.zeros(1, device=DEV)
    reqs = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
    SCALES = ((1.0, "trace prompt lengths + 100"), (0.5, "half of them")) if "--small" not in sys.argv else ((0.25, "a quarter of them"), (0.12, "an eighth of them"))
    for scale, what in SCALES:
        lens = [int(p * scale) + 100 for p, _ in reqs]
        B, ctx = len(lens), max(lens) + 8
        kc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        vc = torch.randn(B, ctx, Hkv, D, device=DEV, dtype=torch.float16)
        q = torch.randn(B, 1, Hq, D, device=DEV, dtype=torch.float16)
        kn = torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16)
        vn = torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16)
        out = torch.empty_like(q)
        idx = torch.arange(B, dtype=torch.int32, device=DEV)
        by = sum(2.0 * (l + 1) * Hkv * D * 2 for l in lens) + B * Hq * D * 2 * 2
        print("== %d sequences, %s: mean %d, max %d tokens, %.0f MB per launch" % (B, what, sum(lens) // B, max(lens), by / 1e6))

        def run(cl_t, host, tiles, label, variant=LEGACY, nwg=0):
            cap = []
            f = lambda: flash_attn_with_kvcache(q, kc[:, :max(lens) + 1], vc[:, :max(lens) + 1], kn, vn, cache_seqlens=cl_t, cache_batch_idx=idx, causal=True, out=out,
                                                _cache_seqlens_host=host, _plan_tiles=tiles, _params_out=cap, _variant=variant, num_splits=nwg)
            ms = timeit(f)
            print("  %-46s %.4f ms  %6.0f GB/s (%.1f%% of 8000)  items %d" % (label, ms, by / ms / 1e6, by / ms / 1e6 / 80, cap[-1].num_split_items))

        cl = torch.tensor(lens, dtype=torch.int32, device=DEV)
        run(cl, None, 0, "DEVICE-planned stream (product default)", 0)
        run(cl, None, 0, "device-planned stream, in-launch merge (lab)", 1 << 20)
        for n in (256, 384, 512, 640, 768, 1024, 1536):
            run(cl, None, 0, "device-planned stream, %d workgroups" % n, 0, -n)
        import vattention_amd.flash_attn as FA
        for x in (0, 2, 8, 16):
            FA._STREAM_SWITCH = x + 1
            run(cl, None, 0, "device-planned stream, switch allowance %d tiles" % x, 0)
        FA._STREAM_SWITCH = 0
        run(cl, None, 0, "uniform split")
        run(cl, lens, 0, "length-balanced plan (planner's piece length)")
        for t in (24, 32, 48, 64, 96, 128):
            run(cl, lens, t, "length-balanced plan, %d tiles per piece" % t)
        eq = [sum(lens) // B] * B
        cle = torch.tensor(eq, dtype=torch.int32, device=DEV)
        by = sum(2.0 * (l + 1) * Hkv * D * 2 for l in eq) + B * Hq * D * 2 * 2
        lens = eq
        run(cle, None, 0, "EQUAL lengths (the mean), uniform split")
        run(cle, None, 0, "EQUAL lengths (the mean), device-planned stream", 0)


if __name__ == "__main__":
    main()
