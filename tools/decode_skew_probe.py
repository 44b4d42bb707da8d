#!/usr/bin/env python3
"""Where do the ~15 us that a decode call spends beyond its stream go?  LAB build: every workgroup of decode_stream_kernel stamps the
100 MHz wall clock at entry, after the plan prologue and at exit (csrc/decode_body.h, `ts`); this tool launches one shape over rotating
caches and prints the distribution of start / plan / finish times relative to the first workgroup's entry, per XCD.
usage: python tools/decode_skew_probe.py [B ctx Hq Hkv [splits]]   (splits < 0: force the stream decomposition with -splits workgroups per kv head,
e.g. for one sequence, which the product plans with the per-sequence split; then only the product's issue order is probed)"""
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
    splits = int(sys.argv[5]) if len(sys.argv) >= 6 else 0
    for fair in ((0,) if splits else (0, 1)):
        print("== fair-share issue priority %s" % ("ON (lab, variant bit 23)" if fair else "OFF (product)"))
        one(B, ctx, Hq, Hkv, LAB | (fair << 23), splits, int(sys.argv[6]) if len(sys.argv) >= 7 else -1)


def one(B, ctx, Hq, Hkv, variant, splits=0, head_off=-1):
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
        if head_off >= 0:      # the kernel's heads 0..Hkv-1 are heads head_off.. of a cache with twice as many (rows twice as long): does a
            # late head follow its INDEX (dispatch order / CUs) or its BYTES (address bits)?
            kc = torch.randn(B, ctx, 2 * Hkv, 128, device=DEV, dtype=torch.float16)[:, :, head_off:head_off + Hkv]
            vc = torch.randn(B, ctx, 2 * Hkv, 128, device=DEV, dtype=torch.float16)[:, :, head_off:head_off + Hkv]
        else:
            kc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
            vc = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
        ps.append(params(q, kc, vc, cl, idx, kn, vn, variant=variant, splits=splits))
    d = K.describe(ps[0][0], lib)
    nwg = d["workgroups"]
    ts = torch.zeros(4096 + 4 * nwg + 8, dtype=torch.int64, device=DEV)
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
    # placement: which CU (XCC, shader engine, array, CU) each workgroup ran on
    hw = ts[4096 + 3 * nwg:4096 + 4 * nwg].cpu()
    cu = ((hw >> 32) & 15) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
    uniq, cnt = torch.unique(cu, return_counts=True)
    hist = torch.bincount(cnt)
    print("  placement: %d workgroups on %d distinct CUs; CUs holding k workgroups: %s" % (nwg, len(uniq), {int(k): int(v) for k, v in enumerate(hist) if v}))
    per_cu = {int(u): int(c) for u, c in zip(uniq, cnt)}
    mates = torch.tensor([per_cu[int(c)] for c in cu])
    for k in sorted(set(mates.tolist())):
        sel = mates == k
        print("    workgroups on a CU with %d resident: %4d, exit mean %.2f us" % (k, int(sel.sum()), float(t[sel, 2].mean())))
    print("    per XCC: %s workgroups" % torch.bincount(((hw >> 32) & 15)).tolist())
    print("shape B=%d ctx=%d Hq=%d Hkv=%d%s: %d workgroups (%s), decode + merge launch %.1f us by events" % (B, ctx, Hq, Hkv, " (heads %d.. of a %d-head cache)" % (head_off, 2 * Hkv) if head_off >= 0 else "", nwg, d, ms))
    q_ = lambda x, f: float(x.sort().values[min(len(x) - 1, int(f * len(x)))])
    for name, col in (("entry", 0), ("plan done", 1), ("exit", 2)):
        x = t[:, col]
        print("  %-10s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us after the first workgroup's entry" % (name, x.min(), q_(x, 0.1), q_(x, 0.5), q_(x, 0.9), x.max()))
    print("  exit MEAN %.2f vs MAX %.2f us: a perfect balancer of the same work could end the stream %.1f %% earlier" % (t[:, 2].mean(), t[:, 2].max(), 100 * (1 - t[:, 2].mean() / t[:, 2].max())))
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
    if nwg % Hkv == 0:
        eh = t[:, 2].view(Hkv, -1)
        ok = eh > 0
        print("  exit by kv head (mean, max): %s" % [(round(float(eh[h][ok[h]].mean()), 1), round(float(eh[h].max()), 1)) for h in range(Hkv)])
    if B == 1 and splits < 0:
        # one sequence: workgroup id = kv head * n + w, w = position of the workgroup's range in the sequence.  Are the late ones the same
        # POSITIONS under every head (an address effect) or scattered (a CU / channel-timing effect)?
        n = -splits
        e = t[:, 2].view(-1, n)
        print("  exit by kv head (mean): %s" % [round(float(x), 1) for x in e.mean(1)])
        pos = e.mean(0)
        print("  exit by position in the sequence, mean over heads, octiles of the sequence: %s" % [round(float(pos[i * n // 8:(i + 1) * n // 8].mean()), 1) for i in range(8)])
        print("  spread of the per-position means %.2f us (std) vs spread of all workgroups %.2f us (std)" % (float(pos.std()), float(t[:, 2].std())))
        worst = torch.argsort(t[:, 2], descending=True)[:16]
        print("  16 latest workgroups (kv head, position, id %% 8, exit us): %s" % [(int(i) // n, int(i) % n, int(i) % 8, round(float(t[i, 2]), 1)) for i in worst])
        rows2 = [r[0][:, 2] for r in rows]
        if len(rows2) >= 2:
            a, b2 = rows2[-1], rows2[-2]
            c = float(((a - a.mean()) * (b2 - b2.mean())).mean() / (a.std() * b2.std() + 1e-9))
            print("  correlation of per-workgroup exit times between two consecutive launches (other cache tensors, same ids): %.2f" % c)
    for x in range(8):
        sel = torch.arange(nwg) % 8 == x
        print("  XCD %d (workgroup id %% 8): exit p50 %6.2f  max %6.2f us" % (x, q_(t[sel, 2], 0.5), t[sel, 2].max()))


if __name__ == "__main__":
    main()
