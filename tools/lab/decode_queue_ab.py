#!/usr/bin/env python3
"""Round 6, VERDICT r05 next-round item 3: a QUEUE OF FIXED SMALL PIECES for decode — the resident workgroups draw pieces of P tiles from a
device counter (tools/lab/csrc/decode_body_lab.h, variant bit 21; P in split_reserved bits 8-15) instead of each streaming ONE range of
total / nwg positions; partials merged by the same second launch.  Against the product's stream plan on the dynamic legs' ragged batches
(where decode is 62-80 % of GPU time) and on the static B16 @ 32 k, over rotating caches, interleaved on one box.
usage: python tools/lab/decode_queue_ab.py [rounds]"""
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Q = 1 << 21
FORMS = ((0, 0, "product: one range per workgroup (stream plan)"), (Q, 8, "drawn queue, pieces of 8 tiles (256 keys)"), (Q, 16, "drawn queue, pieces of 16 tiles (512 keys)"),
         (Q, 32, "drawn queue, pieces of 32 tiles (1 k keys)"), (Q, 64, "drawn queue, pieces of 64 tiles (2 k keys)"))


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
    yield "ragged 200 seqs, trace lengths, llama-3-8b 32/8 heads", 32, 8, trace[:200]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.zeros(1, device=DEV)
    torch.manual_seed(0)
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
        def mk(v, P, kc, vc):
            p, keep = params(q, kc[:, :max(lens) + 1], vc[:, :max(lens) + 1], cl, idx, kn, vn, variant=v)
            if P:      # piece length of the drawn queue (the workspace was sized for the default 16: re-size)
                p.split_reserved |= P << 8
                need = lib.vattn_attn_workspace_bytes(C.byref(p))
                w = torch.empty(need // 4 + 1, dtype=torch.float32, device=DEV)
                p.workspace = w.data_ptr()
                keep.append(w)
            return p, keep
        FK = [(v, P) for v, P, _ in FORMS]
        ps = {k: [mk(k[0], k[1], kc, vc) for kc, vc in caches] for k in FK}
        for v in FK:          # warm-up, and the forms against each other (keep[0] is the launch's output tensor)
            for _rep in range(2):
                for p, _k in ps[v]:
                    assert lib.vattn_flash_attn_with_kvcache(C.byref(p), st) == 0, K.last_error(lib)
            torch.cuda.synchronize()
        diffs = {v: max(float((ps[v][c][1][0].float() - ps[(0, 0)][c][1][0].float()).abs().max()) for c in range(R)) for v in FK}
        res = {v: [] for v in FK}
        iters = max(3, 60 // R)
        for _ in range(rounds):
            for v in FK:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _i in range(iters):
                    for p, _k in ps[v]:
                        lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) * 1e3 / (iters * R))
        d = K.describe(ps[(0, 0)][0][0], lib)
        print("== %s: %d sequences, mean %d tokens, %.0f MB per launch, %d workgroups, %d caches in rotation" % (name, B, sum(lens) // B, by / 1e6, d["workgroups"], R))
        base = statistics.median(res[(0, 0)])
        for v0, P, label in FORMS:
            v = (v0, P)
            m = statistics.median(res[v])
            print("  %-58s median %8.1f us (min %8.1f)  %6.0f GB/s = %.3f of 8 TB/s   x%.3f of the product   |out - product| <= %.1e" % (
                label, m, min(res[v]), by / m / 1e3, by / m / 1e3 / 8000, m / base, diffs[v]))
        del ps, caches


if __name__ == "__main__":
    main()
