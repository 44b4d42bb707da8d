#!/usr/bin/env python3
"""LAB (round 6): where the tile step of prefill64_kernel spends its cycles.

The lab build behind variant bits 28-30 = 3 (the product schedule of round 6: V^T fragments pre-read; round-6 call 2 also stamped
round 5's schedule, profiles/r06_p64_stamps.txt) stamps s_memtime at every 8th MFMA group
(8 stamps per tile step: groups 0, 8, 16, 24 of phase A = S'(t+1) = K.Q^T, groups 0, 8, 16, 24 of phase B = O += V^T.P^T) for the first
64 steps of every wave of every workgroup.  This reads the stamps back and prints, for the steady-state steps of the long workgroups,
the mean cycles of each of the 8 segments, per wave — against 8 x 32 = 256 cycles of matrix pipe per segment — plus the skew between
the four waves of a workgroup at the step's entry.  (Stamping costs: the stamped launch is timed beside the unstamped one.)
usage: python tools/lab/p64_stamps.py [n_tokens]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from tools.kbench import params, time_ms  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32702
    Hq, Hkv = 32, 4
    torch.manual_seed(0)
    q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
    kc = torch.randn(1, n, Hkv, 128, device=DEV, dtype=torch.float16)
    vc = torch.randn(1, n, Hkv, 128, device=DEV, dtype=torch.float16)
    cl = torch.tensor([n], dtype=torch.int32, device=DEV)
    nwg = ((n + 255) // 256) * Hq
    fl = 4.0 * Hq * 128 * (n * (n + 1) / 2)
    for sel in (3,):
        v = 14 | (sel << 28)
        p, keep = params(q, kc, vc, cl, variant=v)
        lse_words = (Hq * n + 1) // 2                                        # the launch's own LSE rows (fp32), in 8-byte words
        raw = torch.zeros(lse_words + nwg * 1024, dtype=torch.int64, device=DEV)
        buf = raw[lse_words:].view(torch.int32)                              # 2048 u32 stamps per workgroup
        p.softmax_lse = raw.data_ptr()
        lib = K.klib_for(v)
        rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(K.last_error(lib))
        torch.cuda.synchronize()
        ms_st = time_ms(p, 1, 3)
        p0, keep0 = params(q, kc, vc, cl, variant=14 | (1 << 28))
        ms_0 = time_ms(p0, 1, 3)
        t = buf.view(nwg, 4, 64, 8).cpu().long()
        t = torch.where(t < 0, t + (1 << 32), t)                             # u32
        full = (t[:, :, 63, 7] != 0) & (t[:, :, 0, 0] != 0)                  # waves that ran at least 64 steps
        # unwrap the 32-bit clock along (step, stamp) of each wave
        flat = t.view(nwg, 4, 512)
        d = flat[:, :, 1:] - flat[:, :, :-1]
        d = torch.where(d < -(1 << 31), d + (1 << 32), d)
        flat = torch.cat([flat[:, :, :1], flat[:, :, :1] + d.cumsum(-1)], -1)
        # the four waves of a workgroup share the counter: re-base waves 1-3 on wave 0 modulo 2^32
        off = flat[:, :, 0] - flat[:, :1, 0]
        off = torch.where(off > (1 << 31), off - (1 << 32), torch.where(off < -(1 << 31), off + (1 << 32), off))
        flat = flat - flat[:, :, :1] + flat[:, :1, :1] + off.unsqueeze(-1)
        t = flat.view(nwg, 4, 64, 8).double()
        wgs = full.all(dim=1).nonzero().flatten()
        print("== build %d (%s): stamped %.3f ms, unstamped %.3f ms (%.0f TF); %d of %d workgroups ran >= 64 steps" % (
            sel, "V^T fragments pre-read in phase A's tail", ms_st, ms_0, fl / ms_0 / 1e9, len(wgs), nwg))
        if not len(wgs):
            continue
        ts = t[wgs]                                                          # [W, 4, 64, 8]
        # segment k of step s: stamp k+1 - stamp k (k = 7: next step's stamp 0); steps 8..62 (steady state, away from the prologue)
        seg = torch.empty(len(wgs), 4, 54, 8, dtype=torch.float64)
        for k in range(7):
            seg[..., k] = ts[:, :, 8:62, k + 1] - ts[:, :, 8:62, k]
        seg[..., 7] = ts[:, :, 9:63, 0] - ts[:, :, 8:62, 7]
        step_cyc = seg.sum(-1)
        names = ["A0-7", "A8-15", "A16-23", "A24-31", "B0-7", "B8-15", "B16-23", "B24-31"]
        print("   cycles per tile step (64 MFMAs = 2048 of matrix pipe): mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f" % (
            step_cyc.mean(), step_cyc.flatten().quantile(0.1), step_cyc.flatten().quantile(0.5), step_cyc.flatten().quantile(0.9)))
        print("   segment      " + "  ".join("%7s" % x for x in names))
        for w in range(4):
            print("   wave %d  mean " % w + "  ".join("%7.0f" % seg[:, w, :, k].mean() for k in range(8)))
        print("   all     mean " + "  ".join("%7.0f" % seg[..., k].mean() for k in range(8)))
        print("   all     p90  " + "  ".join("%7.0f" % seg[..., k].flatten().quantile(0.9) for k in range(8)))
        print("   over 256:    " + "  ".join("%7.0f" % (seg[..., k].mean() - 256) for k in range(8)) + "   (sum %.0f)" % (step_cyc.mean() - 2048))
        # skew of the four waves at step entry (stamp 0) and behind the barrier's segment (stamp 5 = group 8 of phase B is the barrier group)
        e = ts[:, :, 8:62, 0]
        print("   wave skew at step entry: max - min over the 4 waves: mean %.0f cycles, p90 %.0f" % ((e.max(1).values - e.min(1).values).mean(), (e.max(1).values - e.min(1).values).flatten().quantile(0.9)))
        b = ts[:, :, 8:62, 6]
        print("   wave skew at phase-B group 16 (behind the barrier): mean %.0f, p90 %.0f" % ((b.max(1).values - b.min(1).values).mean(), (b.max(1).values - b.min(1).values).flatten().quantile(0.9)))


if __name__ == "__main__":
    main()
