#!/usr/bin/env python3
"""Round 5, VERDICT r04 next-round item 5: do XCD-consecutive ranges with a hand-over inside the XCD's L2 beat the second launch?
LAB build (csrc/decode_body.h, decode_stream_kernel), four forms of the device-planned stream decode on the same shapes, interleaved:
  product            consecutive ranges on consecutive XCDs, records merged by decode_stream_combine_kernel (a second launch)
  bit 20             ... merged inside the launch by the last finisher: write-through records, device-scope ticket  (round 4: equal or slower)
  bit 27             the ranges of ONE XCD are consecutive (a sequence's pieces share an L2), second-launch merge  (what the re-mapping alone costs)
  bits 27 + 20       ... and a sequence inside one XCD is handed over in that L2: write-back records, L2 ticket, loads that only skip the vL1D
First the premise is checked with the lab's placement stamps (variant bit 22): workgroup ids that are equal modulo 8 run on the same XCC.
usage: python tools/decode_xcd_probe.py [rounds]"""
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = ((0, "product (second-launch merge)"), (1 << 20, "in-launch merge, device scope (bit 20)"), (1 << 27, "XCD-consecutive ranges, second launch (bit 27)"),
         ((1 << 27) | (1 << 20), "XCD-consecutive ranges, merge in the XCD's L2 (27+20)"))


def placement():
    B, ctx, Hq, Hkv = 16, 32768, 32, 4
    q = torch.randn(B, 1, Hq, 128, device=DEV, dtype=torch.float16)
    kc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
    vc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
    cl = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
    idx = torch.arange(B, dtype=torch.int32, device=DEV)
    lib = K.klib_lab()
    st = torch.cuda.current_stream().cuda_stream
    p, keep = params(q, kc, vc, cl, idx, None, None, variant=(1 << 22) | (1 << 27))
    d = K.describe(p, lib)
    nwg = d["workgroups"]
    ts = torch.zeros(4096 + 4 * nwg + 8, dtype=torch.int64, device=DEV)
    p.softmax_lse = ts.data_ptr()
    for _ in range(3):
        ts.zero_()
        assert lib.vattn_flash_attn_with_kvcache(C.byref(p), st) == 0, K.last_error(lib)
        torch.cuda.synchronize()
    hw = ts[4096 + 3 * nwg:4096 + 4 * nwg].cpu()
    xcc = ((hw >> 32) & 15).tolist()
    per_res = [sorted(set(xcc[i] for i in range(r, nwg, 8))) for r in range(8)]
    ok = all(len(s) == 1 for s in per_res) and len(set(s[0] for s in per_res)) == 8
    print("placement, %d workgroups (%d per kv head): XCCs seen by workgroup id mod 8 = %s -> premise %s" % (nwg, nwg // Hkv, per_res, "HOLDS" if ok else "DOES NOT HOLD"))
    del keep


def shapes():
    reqs = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_arxiv_lengths_256.json")))["requests"]
    trace = [int(pl) + 100 for pl, _ in reqs]
    yield "ragged 256 seqs, trace lengths, TP8 rank 8/1 heads", 8, 1, trace
    yield "ragged 256 seqs, quarter lengths, TP8 rank 8/1 heads", 8, 1, [l // 4 + 100 for l in trace]
    yield "ragged 64 seqs, trace lengths, llama-3-8b 32/8 heads", 32, 8, trace[:64]
    yield "ragged 48 seqs, half lengths, yi-6b 32/4 heads", 32, 4, [l // 2 + 100 for l in trace[64:112]]
    yield "B16 @ 32k yi-6b 32/4 heads", 32, 4, [32767] * 16
    yield "B64 @ 8k llama-3-8b 32/8 heads", 32, 8, [8191] * 64
    yield "B4 @ 32k yi-6b 32/4 heads", 32, 4, [32767] * 4


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.zeros(1, device=DEV)
    torch.manual_seed(0)
    placement()
    lib = K.klib_lab()
    st = torch.cuda.current_stream().cuda_stream
    for name, Hq, Hkv, lens in shapes():
        B, ctx = len(lens), max(lens) + 8
        by = sum(2.0 * (l + 1) * Hkv * 128 * 2 for l in lens) + B * Hq * 128 * 2 * 2
        R = max(1, int(1.2e9 // by) + 1)          # rotate over caches until a round of launches exceeds the 256 MiB Infinity Cache several times
        q = torch.randn(B, 1, Hq, 128, device=DEV, dtype=torch.float16)
        kn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
        vn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor(lens, dtype=torch.int32, device=DEV)
        idx = torch.arange(B, dtype=torch.int32, device=DEV)
        caches = [(torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16), torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)) for _ in range(R)]
        ps = {v: [params(q, kc[:, :max(lens) + 1], vc[:, :max(lens) + 1], cl, idx, kn, vn, variant=v) for kc, vc in caches] for v, _ in FORMS}
        for v, _ in FORMS:          # warm-up, and the forms against each other (keep[0] is the launch's output tensor)
            for _rep in range(2):
                for p, _k in ps[v]:
                    assert lib.vattn_flash_attn_with_kvcache(C.byref(p), st) == 0, K.last_error(lib)
            torch.cuda.synchronize()
        diffs = {v: max(float((ps[v][c][1][0].float() - ps[0][c][1][0].float()).abs().max()) for c in range(R)) for v, _ in FORMS}
        res = {v: [] for v, _ in FORMS}
        iters = max(3, 60 // R)
        for _ in range(rounds):
            for v, _ in FORMS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _i in range(iters):
                    for p, _k in ps[v]:
                        lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) * 1e3 / (iters * R))
        d = K.describe(ps[0][0][0], lib)
        print("== %s: %d sequences, mean %d tokens, %.0f MB per launch, %d workgroups, %d caches in rotation" % (name, B, sum(lens) // B, by / 1e6, d["workgroups"], R))
        base = statistics.median(res[0])
        for v, label in FORMS:
            m = statistics.median(res[v])
            print("  %-58s median %8.1f us (min %8.1f)  %6.0f GB/s = %.3f of 8 TB/s   x%.3f of the product   |out - product| <= %.1e" % (
                label, m, min(res[v]), by / m / 1e3, by / m / 1e3 / 8000, m / base, diffs[v]))
        del ps, caches


if __name__ == "__main__":
    main()
