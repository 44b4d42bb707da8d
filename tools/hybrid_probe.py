#!/usr/bin/env python3
"""Hybrid batch (one prefill chunk + a decode batch in the same iteration, Sarathi scheduler): does running the two
attention kernels on two HIP streams overlap the matrix-bound prefill with the HBM-bound decode on MI355X?
Times, per layer: prefill alone (default plan = KV-split when its grid underfills the chip, and single pass), decode alone,
both serial on one stream, both on two non-blocking streams.  usage: python tools/hybrid_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")


def launch(p, stream):
    rc = K.klib().vattn_flash_attn_with_kvcache(C.byref(p), C.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(K.last_error())


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
    s_main = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = [("llama8b chunk1k@15k + B64@16k", 32, 8, 1024, 15360, 64, 16384),
             ("llama8b chunk512@8k + B128@8k", 32, 8, 512, 7680, 128, 8192),
             ("llama8b chunk2k@30k + B32@32k", 32, 8, 2048, 30720, 32, 32768),
             ("yi6b chunk4k@28k + B16@32k", 32, 4, 4096, 28672, 16, 32768),
             ("llama70b/tp8 chunk2k@30k + B64@32k", 8, 1, 2048, 30720, 64, 32768),
             ("llama8b chunk512@4k + B64@4k", 32, 8, 512, 3584, 64, 4096),
             ("llama8b chunk256@8k + B32@8k", 32, 8, 256, 7936, 32, 8192)]
    for name, Hq, Hkv, n, c, B, ctx in cases:
        torch.manual_seed(0)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
        kc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
        qd = torch.randn(B, 1, Hq, 128, device=DEV, dtype=torch.float16)
        kd = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
        vd = torch.randn(B, ctx, Hkv, 128, device=DEV, dtype=torch.float16)
        kn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
        vn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=torch.float16)
        cld = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
        idx = torch.arange(B, dtype=torch.int32, device=DEV)
        pd, keepd = params(qd, kd, vd, cld, idx, kn, vn)
        print("== %s" % name)
        pa, keepa = params(q, kc, vc, cl, splits=0)      # default plan: KV-split when the grid underfills the chip
        pu, keepu = params(q, kc, vc, cl, splits=1)      # single pass
        t_pa = timeit(lambda: launch(pa, s_main))
        t_pu = timeit(lambda: launch(pu, s_main))
        t_d = timeit(lambda: launch(pd, s_main))
        t_ser = timeit(lambda: (launch(pa, s_main), launch(pd, s_main)))

        def both(pp):
            def f():
                s1.wait_stream(s_main)
                s2.wait_stream(s_main)
                launch(pd, s2)
                launch(pp, s1)
                s_main.wait_stream(s1)
                s_main.wait_stream(s2)
            return f

        t_par_u = timeit(both(pu))
        t_par_a = timeit(both(pa))
        # the fused launch (vattn_hybrid_attn of the LAB library; the product's entry point is the serial order): by-arrival roles, and with
        # every workgroup preferring one queue (A/B)
        lib = K.klib_lab()
        need = lib.vattn_hybrid_workspace_bytes(C.byref(pu), C.byref(pd))
        ws = torch.zeros(need // 4 + 64, dtype=torch.float32, device=DEV)

        def fused(mode):
            def f():
                pu.variant = (pu.variant & ~(3 << 12)) | (mode << 12)
                rc = lib.vattn_hybrid_attn(C.byref(pu), C.byref(pd), C.c_void_p(ws.data_ptr()), C.c_void_p(s_main.cuda_stream))
                if rc != 0:
                    raise RuntimeError(K.last_error(lib))
            return f

        t_f0, t_f1, t_f2 = timeit(fused(0)), timeit(fused(1)), timeit(fused(2))
        pd.variant |= 1024                    # split merge through device-scope accesses instead of fences
        t_f3 = timeit(fused(0))
        pd.variant &= ~1024
        pu.variant &= ~(3 << 12)
        print("  prefill %.3f ms (single pass %.3f)  decode %.3f ms | serial, default plan %.3f ms | two streams: single-pass prefill %.3f ms, "
              "default-plan prefill %.3f ms | FUSED launch %.3f ms (prefill-first roles %.3f, decode-first roles %.3f; fence-free merge %.3f) | streams/serial %.2fx  fused/serial %.2fx"
              % (t_pa, t_pu, t_d, t_ser, t_par_u, t_par_a, t_f0, t_f1, t_f2, t_f3, t_ser / min(t_par_u, t_par_a, t_ser), t_ser / min(t_f0, t_f3)))
        del keepa, keepu
        del keepd, kd, vd


if __name__ == "__main__":
    main()
