#!/usr/bin/env python3
"""Round 6, VERDICT r05 item 7: is a SINGLE prompt on a TP8 rank (8 query / 1 kv head — what the reference's unmodified wrapper launches,
one call per prompt and layer, vattention_flashattention_wrapper.py:129-174) already launched at its split optimum?  For prompts of the
dynamic trace's lengths: the planner's own choice (work list or default launch) against EVERY forced piece length (work lists of pieces
of at most T key tiles, T = 2 .. 96) and the default grid — same box, interleaved.  "planner / best" = 1.00 means nothing is left to take
by re-planning the reference wrapper's per-prompt calls.   usage: python tools/lab/tp8_single_prompt_sweep.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from tools.kbench import params, time_ms  # noqa: E402
from vattention_amd import flash_attn as FA  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv = 8, 1


def launchable(p, n, force, keep):
    """p with the planner's (force = 0) or a forced work list attached; None when the planner keeps the default launch"""
    pl = FA.prefill_plan(p, [n], [n], DEV, force_tiles=force, persistent=False)
    if pl.t is None:
        return False
    pl.attach(p)
    need = K.klib().vattn_attn_workspace_bytes(C.byref(p))
    w = torch.empty(need // 4 + 1, dtype=torch.float32, device=DEV)
    p.workspace = w.data_ptr()
    keep += [pl, w]
    return True


def main():
    torch.zeros(1, device=DEV)
    print("%-8s %-22s %-10s %-34s %s" % ("prompt", "planner", "default", "best forced piece length", "planner / best"))
    for n in (2048, 3000, 4119, 5000, 6526, 7341, 8192, 9441, 10159, 12000, 14505, 16384, 20751):
        torch.manual_seed(n)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
        kc = torch.randn(1, n, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(1, n, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor([n], dtype=torch.int32, device=DEV)
        arms = {}
        p, keep = params(q, kc, vc, cl, variant=0)
        arms["default"] = (p, keep)
        p, keep = params(q, kc, vc, cl, variant=0)
        planner_list = launchable(p, n, 0, keep)
        arms["planner"] = (p, keep)
        for T in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
            p, keep = params(q, kc, vc, cl, variant=0)
            if launchable(p, n, T, keep):
                arms[T] = (p, keep)
        res = {k: [] for k in arms}
        for _rep in range(3):
            for k, (p, _keep) in arms.items():
                res[k].append(time_ms(p, 2, 10))
        med = {k: sorted(v)[1] for k, v in res.items()}
        forced = {k: v for k, v in med.items() if isinstance(k, int)}
        bt = min(forced, key=forced.get) if forced else None
        best = min(med.values())
        fl = 4.0 * Hq * 128 * n * (n + 1) / 2
        print("%-8d %-22s %-10s %-34s %.3f" % (n, "%.4f ms %s" % (med["planner"], "(list)" if planner_list else "(default)"), "%.4f" % med["default"],
                                              ("T = %d tiles: %.4f ms (%.0f TF)" % (bt, forced[bt], fl / forced[bt] / 1e9)) if bt else "-", med["planner"] / best))


if __name__ == "__main__":
    main()
