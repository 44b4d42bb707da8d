#!/usr/bin/env python3
"""f1, one bounded experiment: does confining the matrix-bound prefill and the HBM-bound decode of a hybrid batch to
DISJOINT sets of CUs (two streams created with hipExtStreamCreateWithCUMask) beat running them one after the other on the whole chip?
Hypothesis under test: a power-capped prefill on fewer CUs clocks higher while decode streams on the rest.

Per hybrid shape (the four of profiles/r02_hybrid_probe.txt) and per layer: prefill alone / decode alone on the whole chip, serial,
two plain non-blocking streams, and two CU-masked streams for several (prefill CUs / decode CUs) partitions:
  192/64 and 128/128 as whole XCDs (6 + 2, 4 + 4), and the same counts taken as a slice of EVERY XCD (24 + 8, 16 + 16 CUs of each).
The CU mask is a bit vector over the chip's CUs; which bit is which physical CU is not documented for this part, so three readings
of each split are measured, each with its kernels ALONE on their partition as the calibration.
usage: python tools/cumask_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.kbench import params  # noqa: E402
from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(bits, ncu):
    words = (ncu + 31) // 32
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask rc=%d" % rc)
    return torch.cuda.ExternalStream(s.value, device=DEV), s


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


def xcd_of_bits(ncu):
    """Assumed bit -> XCD map: the dispatcher's workgroup i -> XCD i % 8 pattern carried over to the mask's bit order.  The map is not
    documented for this part, so three readings of "192/64" and "128/128" are measured (bit % 8 = XCD, a slice of every group of 32
    bits, contiguous bits); the per-partition "alone" timings printed beside them show what each partition actually delivers."""
    return [b % 8 for b in range(ncu)]


def main():
    torch.zeros(1, device=DEV)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    s_main = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    xcd = xcd_of_bits(ncu)
    by_xcd = [[b for b in range(ncu) if xcd[b] == x] for x in range(8)]
    per = ncu // 8
    parts = {
        "192/64 whole XCDs (bit mod 8)": (sum(by_xcd[:6], []), sum(by_xcd[6:], [])),
        "128/128 whole XCDs (bit mod 8)": (sum(by_xcd[:4], []), sum(by_xcd[4:], [])),
        "192/64 slice of every XCD": (sum([l[:per * 3 // 4] for l in by_xcd], []), sum([l[per * 3 // 4:] for l in by_xcd], [])),
        "128/128 slice of every XCD": (sum([l[:per // 2] for l in by_xcd], []), sum([l[per // 2:] for l in by_xcd], [])),
        "192/64 contiguous bits": (list(range(ncu * 3 // 4)), list(range(ncu * 3 // 4, ncu))),
        "128/128 contiguous bits": (list(range(ncu // 2)), list(range(ncu // 2, ncu))),
    }
    streams = {}
    for name, (a, b) in parts.items():
        streams[name] = (masked_stream(a, ncu), masked_stream(b, ncu))
    cases = [("llama8b chunk1k@15k + B64@16k", 32, 8, 1024, 15360, 64, 16384),
             ("llama8b chunk512@8k + B128@8k", 32, 8, 512, 7680, 128, 8192),
             ("yi6b chunk4k@28k + B16@32k", 32, 4, 4096, 28672, 16, 32768),
             ("llama70b/tp8 chunk2k@30k + B64@32k", 8, 1, 2048, 30720, 64, 32768)]
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
        pa, keepa = params(q, kc, vc, cl, splits=0)
        t_p = timeit(lambda: launch(pa, s_main))
        t_d = timeit(lambda: launch(pd, s_main))
        t_ser = timeit(lambda: (launch(pa, s_main), launch(pd, s_main)))

        def both(sa, sb):
            def f():
                sa.wait_stream(s_main)
                sb.wait_stream(s_main)
                launch(pd, sb)
                launch(pa, sa)
                s_main.wait_stream(sa)
                s_main.wait_stream(sb)
            return f

        t_two = timeit(both(s1, s2))
        print("== %s\n  whole chip: prefill %.3f ms, decode %.3f ms, serial %.3f ms, two plain streams %.3f ms (%.2fx of serial)" % (
            name, t_p, t_d, t_ser, t_two, t_ser / t_two))
        for pname, ((sa, _ha), (sb, _hb)) in streams.items():
            # each kernel alone on its partition (the clock / bandwidth it gets there); the timing events sit on the main stream
            t_pa = timeit(lambda: (sa.wait_stream(s_main), launch(pa, sa), s_main.wait_stream(sa)))
            t_db = timeit(lambda: (sb.wait_stream(s_main), launch(pd, sb), s_main.wait_stream(sb)))
            t_m = timeit(both(sa, sb))
            print("  %-32s prefill alone on its CUs %.3f ms, decode alone on its CUs %.3f ms, both %.3f ms (%.2fx of serial)" % (
                pname, t_pa, t_db, t_m, t_ser / t_m))
        del keepa, keepd, kd, vd


if __name__ == "__main__":
    main()
