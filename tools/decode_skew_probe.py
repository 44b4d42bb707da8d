#!/usr/bin/env python3
"""Where do the ~15 us that a decode call spends beyond its stream go?  LAB build: every workgroup of decode_stream_kernel stamps the
100 MHz wall clock at entry, after the plan prologue and at exit (csrc/decode_body.h, `ts`); this tool launches one shape over rotating
caches and prints the distribution of start / plan / finish times relative to the first workgroup's entry, per XCD.
usage: python tools/decode_skew_probe.py [B ctx Hq Hkv]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
LAB = 1 << 22          # lab library + timestamps behind softmax_lse (csrc/decode_body.h)


def main():
    B, ctx, Hq, Hkv = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (16, 32768, 32, 4)
    for fair in (0, 1):
        print("== fair-share issue priority %s" % ("ON (lab, variant bit 23)" if fair else "OFF (product)"))
        one(B, ctx, Hq, Hkv, LAB | (fair << 23))


def one(B, ctx, Hq, Hkv, variant):
    torch.zeros(1, device=DEV)
    torch.manual_seed(0)
    q = torch.randn(B, 1, Hq, 128, device=DEV, dtype=torch.float16)
    kn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
    vn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
    cl = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
    idx = torch.arange(B, dtype=torch.int32, device=DEV)
    lib = K.klib_lab()
    st = torch.cuda.current_stream().cuda_stream
    ps = []
    for _ in range(max(2, int(1.5e9 // (B * 2.0 * ctx * Hkv * 256)) + 1)):
        kc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
        ps.append(params(q, kc, vc, cl, idx, kn, vn, variant=variant))
    d = K.describe(ps[0][0], lib)
    nwg = d["workgroups"]
    ts = torch.zeros(4096 + 3 * nwg + 8, dtype=torch.int64, device=DEV)
    for p, _k in ps:                       # warm-up round without stamps
        lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
    rows = []
    for p, _k in ps:
        p.softmax_lse = ts.data_ptr()
        ts.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), st)
        assert rc == 0, K.last_error(lib)
        e1.record()
        torch.cuda.synchronize()
        t = ts[4096:4096 + 3 * nwg].view(nwg, 3).cpu().double() * 0.01          # 100 MHz ticks -> us
        t0 = t[:, 0].min()
        rows.append((t - t0, e0.elapsed_time(e1) * 1e3))
    t, ms = rows[-1]
    print("shape B=%d ctx=%d Hq=%d Hkv=%d: %d workgroups (%s), decode + merge launch %.1f us by events" % (B, ctx, Hq, Hkv, nwg, d, ms))
    q_ = lambda x, f: float(x.sort().values[min(len(x) - 1, int(f * len(x)))])
    for name, col in (("entry", 0), ("plan done", 1), ("exit", 2)):
        x = t[:, col]
        print("  %-10s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us after the first workgroup's entry" % (name, x.min(), q_(x, 0.1), q_(x, 0.5), q_(x, 0.9), x.max()))
    dur = t[:, 2] - t[:, 0]
    print("  per-workgroup lifetime: min %.2f  p50 %.2f  max %.2f us; plan prologue p50 %.2f us" % (dur.min(), q_(dur, 0.5), dur.max(), q_(t[:, 1] - t[:, 0], 0.5)))
    # by dispatch order: with one round of 768 workgroups on 256 CUs the hardware fills CU slots in id order
    for g in range((nwg + 255) // 256):
        sel = (torch.arange(nwg) // 256) == g
        print("  workgroup ids %4d-%4d: exit p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (256 * g, min(nwg, 256 * g + 256) - 1, q_(t[sel, 2], 0.1), q_(t[sel, 2], 0.5), q_(t[sel, 2], 0.9), t[sel, 2].max()))
    for g in range(6):
        sel = (torch.arange(nwg) // 128) == g
        if sel.any():
            print("  ids //128 == %d: exit p50 %6.2f" % (g, q_(t[sel, 2], 0.5)))
    for x in range(8):
        sel = torch.arange(nwg) % 8 == x
        print("  XCD %d (workgroup id %% 8): exit p50 %6.2f  max %6.2f us" % (x, q_(t[sel, 2], 0.5), t[sel, 2].max()))


if __name__ == "__main__":
    main()
