#!/usr/bin/env python3
"""LAB A/B of STRIPED decode pieces (csrc/decode_body.h, variant bit 24): the pieces of a sequence interleave tile by tile instead of each
streaming a contiguous range.  (1) results against the contiguous form on the same inputs (same kernel, other grouping of the partial
softmaxes: 1e-3), incl. the appended row; (2) timings over rotating caches, lab library both ways.
usage: python tools/decode_striped_ab.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
A, Bv = (1 << 22) | (1 << 25), (1 << 22) | (1 << 24)      # lab library: contiguous pieces everywhere / striped pieces everywhere


def run(variant, q, kc, vc, cl, idx, kn, vn, splits=0):
    kc2, vc2 = kc.clone(), vc.clone()
    p, keep = params(q, kc2, vc2, cl, idx, kn, vn, variant=variant, splits=splits)
    lib = K.klib_lab()
    rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, K.last_error(lib)
    torch.cuda.synchronize()
    return keep[0].float(), kc2, vc2, K.describe(p, lib)


def check():
    torch.manual_seed(1)
    bad = 0
    for (B, ctx, Hq, Hkv, jitter, splits) in [(1, 4099, 28, 4, 0, 0), (1, 131071, 28, 4, 0, 0), (1, 65, 32, 4, 0, 0), (1, 31, 8, 1, 0, 0), (1, 20000, 14, 2, 0, 0),
                                              (3, 1000, 32, 8, 0, 0), (16, 32763, 32, 4, 0, 0), (2, 77, 8, 1, 0, 0), (5, 33, 32, 4, 0, 0), (4, 9000, 32, 4, 40, 0),
                                              (64, 2000, 8, 1, 30, 0), (8, 5000, 32, 8, 3, 0), (2, 8192, 32, 4, 0, 7), (1, 8191, 32, 4, 0, 24), (6, 300, 64, 1, 0, 0)]:
        for dt in (torch.float16, torch.bfloat16):
            q = torch.randn(B, 1, Hq, 128, device=DEV, dtype=dt)
            kc = torch.randn(B + 1, ctx + 1, Hkv, 128, device=DEV, dtype=dt)
            vc = torch.randn(B + 1, ctx + 1, Hkv, 128, device=DEV, dtype=dt)
            kn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=dt)
            vn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=dt)
            cl = (torch.full((B,), ctx, dtype=torch.int32) - torch.randint(0, jitter + 1, (B,), dtype=torch.int32)).to(DEV)
            idx = (torch.randperm(B + 1)[:B]).to(torch.int32).to(DEV)
            oa, ka, va, da = run(A, q, kc, vc, cl, idx, kn, vn, splits)
            ob, kb, vb, db = run(Bv, q, kc, vc, cl, idx, kn, vn, splits)
            err = (oa - ob).abs().max().item()
            same_cache = torch.equal(ka, kb) and torch.equal(va, vb)
            ok = err <= (2e-3 if dt == torch.float16 else 1.6e-2) and same_cache and torch.isfinite(ob).all().item()
            bad += not ok
            print("  %s B=%3d ctx=%6d Hq=%2d Hkv=%d jitter %2d splits %2d %s: max |striped - contiguous| %.2e, caches equal %s, plan %s" % (
                "ok " if ok else "BAD", B, ctx, Hq, Hkv, jitter, splits, "f16 " if dt == torch.float16 else "bf16", err, same_cache,
                {k: da[k] for k in ("path", "nsplit", "workgroups")}))
    print("parity: %s" % ("all ok" if not bad else "%d BAD" % bad))
    return bad


def weighted():
    # LAB bit 26: one sequence, the kv head whose bytes sit in the slow address quarter gets 5 shares of the workgroups where the others get 4
    from tools import kbench
    Wv = (1 << 22) | (1 << 26)
    torch.manual_seed(2)
    bad = 0
    for (ctx, Hq, Hkv) in [(4099, 28, 4), (131071, 28, 4), (20000, 14, 2), (65, 32, 4), (40000, 32, 8), (8191, 8, 1)]:
        for dt in (torch.float16, torch.bfloat16):
            q = torch.randn(1, 1, Hq, 128, device=DEV, dtype=dt)
            kc = torch.randn(2, ctx + 1, Hkv, 128, device=DEV, dtype=dt)
            vc = torch.randn(2, ctx + 1, Hkv, 128, device=DEV, dtype=dt)
            kn = torch.randn(1, 1, Hkv, 128, device=DEV, dtype=dt)
            vn = torch.randn(1, 1, Hkv, 128, device=DEV, dtype=dt)
            cl = torch.full((1,), ctx, dtype=torch.int32, device=DEV)
            idx = torch.ones(1, dtype=torch.int32, device=DEV)
            oa, ka, va, da = run(1 << 22, q, kc, vc, cl, idx, kn, vn)
            ob, kb, vb, db = run(Wv, q, kc, vc, cl, idx, kn, vn)
            err = (oa - ob).abs().max().item()
            ok = err <= (2e-3 if dt == torch.float16 else 1.6e-2) and torch.equal(ka, kb) and torch.equal(va, vb) and torch.isfinite(ob).all().item()
            bad += not ok
            print("  %s weighted heads: ctx=%6d Hq=%2d Hkv=%d %s: max |weighted - even| %.2e" % ("ok " if ok else "BAD", ctx, Hq, Hkv, "f16 " if dt == torch.float16 else "bf16", err))
    print("weighted parity: %s" % ("all ok" if not bad else "%d BAD" % bad))
    if bad:
        return
    kbench.ROTATE = True
    kbench.SPLITS = [0]
    kbench.ONLY = "yi34b/tp2 B1@128k,yi34b/tp4 B1@128k,yi6b B1@32k"
    for rep in range(2):
        for v, name in ((1 << 22, "even split (striped)"), (Wv, "WEIGHTED heads (striped)")):
            print("== %s (lab library), pass %d" % (name, rep + 1))
            kbench.decode(v)


def timing():
    from tools import kbench
    kbench.ROTATE = True
    kbench.SPLITS = [0]
    kbench.ONLY = "yi34b/tp2 B1@128k,yi34b/tp4 B1@128k,yi6b B1@32k,yi6b B2@32k,yi6b B4@32k,yi6b B16@32k,yi34b/tp2 B8@128k,llama70b/tp8 B64@32k,llama8b B64@8k"
    for rep in range(2):
        for v, name in ((A, "contiguous pieces"), (Bv, "STRIPED pieces")):
            print("== %s (lab library), pass %d" % (name, rep + 1))
            kbench.decode(v)


if __name__ == "__main__":
    torch.zeros(1, device=DEV)
    if "--weighted" in sys.argv:
        weighted()
        sys.exit(0)
    bad = check()
    if not bad:
        timing()
