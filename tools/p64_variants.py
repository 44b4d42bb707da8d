#!/usr/bin/env python3
"""Schedule alternatives of prefill64_kernel (lab builds, variant bits 8-11 = 11..15) against the product build (variant 14): results
must be bit-identical (same arithmetic in the same order, only the issue order differs), then the timings of tools/kbench.py.
usage: python tools/p64_variants.py [sel,sel,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params, time_ms  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")


def run(p):
    lib = K.klib_for(p.variant)
    rc = lib.vattn_flash_attn_with_kvcache(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(K.last_error(lib))
    torch.cuda.synchronize()


def main():
    # a build is named by its number in variant bits 8-11, or "r6:N" for the round-6 experiments in bits 28-30
    sels = [x if x.startswith("r6:") else int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [11, 12, 13, 14, 15]
    var = lambda sel: 14 | ((int(sel[3:]) << 28) if isinstance(sel, str) else (sel << 8))
    torch.zeros(1, device=DEV)
    for n, c, Hq, Hkv in [(3000, 500, 8, 2), (777, 1000, 4, 4)]:
        torch.manual_seed(n)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
        kc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
        p0, keep0 = params(q, kc, vc, cl, variant=14)
        run(p0)
        ref = keep0[0].clone()
        for sel in sels:
            p, keep = params(q, kc, vc, cl, variant=var(sel))
            run(p)
            same = torch.equal(keep[0], ref)
            err = (keep[0].float() - ref.float()).abs().max().item()
            print("n=%d c=%d Hq=%d/%d build %s: %s (max |diff| %.3e)" % (n, c, Hq, Hkv, sel, "bit-identical to the product build" if same else "DIFFERS", err))
    for name, Hq, Hkv, n, c in [("yi6b whole", 32, 4, 32702, 0), ("yi6b chunk4k@28k", 32, 4, 4096, 28672), ("llama8b 16k", 32, 8, 16384, 0)]:
        torch.manual_seed(0)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
        kc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
        fl = 4.0 * Hq * 128 * (n * c + n * (n + 1) / 2)
        line = []
        for rep in range(2):
            for sel in [0] + sels:
                p, keep = params(q, kc, vc, cl, variant=var(sel))
                ms = time_ms(p, warmup=2, iters=5)
                line.append("%s: %.3f ms (%.0f TF)" % (sel, ms, fl / ms / 1e9))
        print("%-18s %s" % (name, " | ".join(line)))


if __name__ == "__main__":
    main()
