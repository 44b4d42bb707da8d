#!/usr/bin/env python3
"""Does the decode kernel lose bandwidth over MEGACACHE views backed by HIP VMM pages?  A layer's view k[:, :, l] of a megacache tensor
[B, ctx, L, kvh, D] touches 1/L of every page a sequence owns: with L = 80 a launch walks 80 x more virtual address space (and
translations) per useful byte than a per-layer tensor does.  Times the batch-64 @ 32 k decode of one TP=8 rank of Llama-3-70B (8 / 1
heads) over (a) a torch allocation with the megacache stride, (b) the page manager's virtual tensors with 2 / 8 / 32 / 128 MiB pages.
usage: python tools/megacache_tlb_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params, time_ms  # noqa: E402
from vattention_amd import vattention  # noqa: E402

DEV = torch.device("cuda:0")
L, Hq, Hkv, D = 80, 8, 1, 128


def run(B, ctx, kc, vc, what):
    q = torch.randn(B, 1, Hq, D, device=DEV, dtype=torch.float16)
    kn = torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16)
    vn = torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16)
    cl = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
    idx = torch.arange(B, dtype=torch.int32, device=DEV)
    p, keep = params(q, kc, vc, cl, idx, kn, vn)
    ms = time_ms(p, 3, 20)
    by = B * 2.0 * ctx * Hkv * D * 2 + B * Hq * D * 2 * 2
    print("  %-44s B=%3d ctx=%6d : %8.4f ms  %7.1f GB/s  (%.1f%% of 8000)" % (what, B, ctx, ms, by / ms / 1e6, by / ms / 1e6 / 80))


def main():
    torch.zeros(1, device=DEV)
    for B, ctx in ((64, 32768), (256, 6144)):
        print("== decode over one layer's view of a %d-layer megacache, B = %d, context %d" % (L, B, ctx))
        kt = torch.randn(B, ctx, L, Hkv, D, device=DEV, dtype=torch.float16)
        vt = torch.randn(B, ctx, L, Hkv, D, device=DEV, dtype=torch.float16)
        run(B, ctx, kt[:, :, 40], vt[:, :, 40], "torch allocation, megacache stride")
        del kt, vt
        torch.cuda.empty_cache()
        for page_mib in (2, 8, 32, 128):
            page = page_mib << 20
            vattention.enable_layered_async(False)
            ts = vattention.init_kvcache(L, Hkv, D, B, ctx, 0, torch.float16, page, True)
            try:
                tpp = vattention.layout()["tokens_per_page"]
                need = B * ((ctx + tpp - 1) // tpp) * page * 2
                vattention.reserve_physical_pages(need + 4 * 2 * L * page)
                vattention.step([ctx] * B, False)
                k, v = ts[0], ts[1]
                k[:, :, 40].normal_()
                v[:, :, 40].normal_()
                run(B, ctx, k[:, :, 40], v[:, :, 40], "page manager, %d MiB pages (%d handles)" % (page_mib, vattention.stats()["handles_created"]))
            finally:
                vattention.cleanup()


if __name__ == "__main__":
    main()
