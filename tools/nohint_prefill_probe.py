#!/usr/bin/env python3
"""Prefill launches WITHOUT host-side hints: a chunk on a long prefix on one TP = 8 rank of Llama-3-70B (8 / 1 heads),
called the way the reference's wrapper calls it — the slot's whole row-block (max_ctx rows) and `cache_seqlens` as a device tensor
(vattention_flashattention_wrapper.py:146-166) — in three situations:
  hinted    the caller passes _max_seqlen_k (what this package's wrapper does)
  foreign   plain torch tensors, no hint: the library falls back to the view's row count (FlashAttention's rule)
  manager   the tensors are the page manager's and the engine stepped the lengths: the drop-in resolves pointer -> slot -> length
usage: python tools/nohint_prefill_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd import vattention as va  # noqa: E402
from vattention_amd.flash_attn import flash_attn_with_kvcache  # noqa: E402
import vattention_amd.flash_attn as FA  # noqa: E402

DEV = torch.device("cuda:0")
Hq, Hkv, D, CTX = 8, 1, 128, 32768


def timeit(fn, iters=20):
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
    va.enable_layered_async(False)
    tensors = va.init_kvcache(2, Hkv, D, 4, CTX, 0, torch.float16, 2 << 20, False)
    va.reserve_physical_pages(4 << 30)
    try:
        for n, c in ((2048, 30720), (512, 15872), (1024, 7168), (4096, 0)):
            lens = [0, c + n, 0, 0]
            va.step(lens, False)
            k_m, v_m = tensors[0], tensors[2]
            k_m[1, :c + n].normal_()
            v_m[1, :c + n].normal_()
            kf = torch.randn(1, CTX, Hkv, D, device=DEV, dtype=torch.float16)       # a foreign cache with max_ctx rows
            vf = torch.randn(1, CTX, Hkv, D, device=DEV, dtype=torch.float16)
            q = torch.randn(1, n, Hq, D, device=DEV, dtype=torch.float16)
            cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
            out = torch.empty_like(q)
            fl = 4.0 * Hq * D * (n * c + n * (n + 1) / 2)
            row = []
            for name, f in (("hinted", lambda: flash_attn_with_kvcache(q, kf, vf, cache_seqlens=cl, causal=True, out=out, _max_seqlen_k=c + n, _pf_plan="host", _cache_seqlens_host=[c + n])),
                            ("foreign", lambda: flash_attn_with_kvcache(q, kf, vf, cache_seqlens=cl, causal=True, out=out)),
                            ("manager", lambda: flash_attn_with_kvcache(q, k_m[1].unsqueeze(0), v_m[1].unsqueeze(0), cache_seqlens=cl, causal=True, out=out))):
                c0 = dict(FA.counters)
                ms = timeit(f)
                d = {k: FA.counters[k] - c0[k] for k in c0}
                row.append("%s %.4f ms %6.0f TF (pm lengths %d, lists %d)" % (name, ms, fl / ms / 1e9, d["lengths_from_page_manager"], d["work_list_attached"]))
            print("chunk %5d @ %5d: " % (n, c) + " | ".join(row))
    finally:
        va.cleanup()


if __name__ == "__main__":
    main()
