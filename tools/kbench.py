#!/usr/bin/env python3
"""Kernel microbenchmark (GPU): times the attention kernels through the C ABI with HIP events
(vattn_time_attn) on the shapes of BASELINE.md §3 and prints TFLOP/s / GB/s against the rooflines.
usage: python tools/kbench.py [prefill] [decode] [--variant N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")


def params(q, kc, vc, cl, idx=None, kn=None, vn=None, causal=True, splits=0, variant=0, ws=None):
    out = torch.empty_like(q)
    p = K.AttnParams()
    p.q, p.out = q.data_ptr(), out.data_ptr()
    p.q_batch_stride, p.q_row_stride, p.q_head_stride = q.stride(0), q.stride(1), q.stride(2)
    p.o_batch_stride, p.o_row_stride, p.o_head_stride = out.stride(0), out.stride(1), out.stride(2)
    p.k_cache, p.v_cache = kc.data_ptr(), vc.data_ptr()
    p.k_batch_stride, p.k_row_stride, p.k_head_stride = kc.stride(0), kc.stride(1), kc.stride(2)
    p.v_batch_stride, p.v_row_stride, p.v_head_stride = vc.stride(0), vc.stride(1), vc.stride(2)
    if kn is not None:
        p.k_new, p.v_new = kn.data_ptr(), vn.data_ptr()
        p.knew_batch_stride, p.knew_row_stride, p.knew_head_stride = kn.stride(0), kn.stride(1), kn.stride(2)
        p.vnew_batch_stride, p.vnew_row_stride, p.vnew_head_stride = vn.stride(0), vn.stride(1), vn.stride(2)
        p.seqlen_knew = kn.shape[1]
    p.cache_seqlens = cl.data_ptr()
    p.cache_batch_idx = idx.data_ptr() if idx is not None else None
    p.b, p.seqlen_q, p.h, p.d = q.shape[0], q.shape[1], q.shape[2], q.shape[3]
    p.seqlen_k, p.h_k = kc.shape[1], kc.shape[2]
    p.is_causal, p.dtype, p.num_splits, p.softmax_scale, p.variant = int(causal), (1 if q.dtype == torch.bfloat16 else 0), splits, q.shape[3] ** -0.5, variant
    p.max_seqlen_k_hint = kc.shape[1]        # the benchmark caches are exactly as long as the sequences
    p.split_reserved = int(os.environ.get("KBENCH_SWITCH_TILES", "-1")) + 1      # stream decode: switch allowance override (tuning)
    p.split_reserved |= int(os.environ.get("KBENCH_LAB_SUB", "0")) << 16         # lab builds: sub-selector (prefill64 round-6 price list)
    keep = [out, q, kc, vc, cl, idx, kn, vn]
    need = K.klib_for(p.variant).vattn_attn_workspace_bytes(C.byref(p))
    if need:
        w = torch.empty(need // 4 + 1, dtype=torch.float32, device=DEV)
        p.workspace = w.data_ptr()
        keep.append(w)
    return p, keep


def time_ms(p, warmup=2, iters=5):
    lib = K.klib_for(p.variant)       # product library unless the variant names a lab build
    ms = lib.vattn_time_attn(C.byref(p), torch.cuda.current_stream().cuda_stream, warmup, iters)
    if ms < 0:
        raise RuntimeError(K.last_error(lib))
    return ms


def prefill(variant):
    print("== prefill (causal chunk n against c cached), %s, D=128 ==" % ("bf16" if DTYPE == torch.bfloat16 else "fp16"))
    for name, Hq, Hkv, n, c in [("yi6b whole", 32, 4, 32702, 0), ("yi6b chunk4k@28k", 32, 4, 4096, 28672), ("yi6b chunk4k@0", 32, 4, 4096, 0),
                                ("llama8b 16k", 32, 8, 16384, 0), ("yi34b/tp2 chunk16k@112k", 28, 4, 16384, 114688),
                                ("llama70b/tp8 8k", 8, 1, 8192, 0), ("small 2k", 32, 4, 2048, 0),
                                ("llama70b/tp8 chunk2k@30k", 8, 1, 2048, 30720), ("llama70b/tp8 chunk512@16k", 8, 1, 512, 15872),
                                ("yi34b/tp2 chunk1k@64k", 28, 4, 1024, 64512), ("llama70b/tp8 4k", 8, 1, 4096, 0),
                                ("llama70b/tp8 2k", 8, 1, 2048, 0), ("llama8b chunk512@8k", 32, 8, 512, 7680)]:
        if ONLY and not any(o in name for o in ONLY.split(",")):
            continue
        torch.manual_seed(0)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=DTYPE)
        kc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=DTYPE)
        vc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=DTYPE)
        scale = float(os.environ.get("KBENCH_DATA_SCALE", "1"))     # 0 = zero-filled inputs (data-dependent power: clocks rise)
        if scale != 1.0:
            q, kc, vc = q * scale, kc * scale, vc * scale
        if os.environ.get("KBENCH_DATA_STYLE") == "same_token":
            # what the reference's own benchmark feeds its kernels: prompt ids [1]*n through random-init weights
            # (scripts/benchmark_e2e_static_trace.py) -> every position carries the SAME hidden state, so v rows are identical and
            # q / k rows are RoPE rotations of one vector per head
            def rope(x, pos0):
                T, H, Dh = x.shape[1], x.shape[2], x.shape[3]
                pos = torch.arange(pos0, pos0 + T, device=DEV, dtype=torch.float32)[:, None]
                inv = 10000.0 ** (-torch.arange(0, Dh, 2, device=DEV, dtype=torch.float32) / Dh)
                ang = pos * inv[None, :]
                cos, sin = ang.cos()[None, :, None, :], ang.sin()[None, :, None, :]
                x = x.float()
                x1, x2 = x[..., 0::2], x[..., 1::2]
                o = torch.empty_like(x)
                o[..., 0::2] = x1 * cos - x2 * sin
                o[..., 1::2] = x1 * sin + x2 * cos
                return o.to(DTYPE)
            q = rope(q[:, :1].expand(-1, n, -1, -1).contiguous(), c)
            kc = rope(kc[:, :1].expand(-1, c + n, -1, -1).contiguous(), 0)
            vc = vc[:, :1].expand(-1, c + n, -1, -1).contiguous()
        cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
        p, keep = params(q, kc, vc, cl, variant=variant, splits=PF_SPLITS)
        tag = ""
        if WORKLIST:          # host-planned work list (vattn_prefill_plan) where the planner wants one
            from vattention_amd import flash_attn as FA
            pl = FA.prefill_plan(p, [n], [c + n], DEV, force_tiles=WL_TILES, persistent=PERSIST, drawn=DRAWN)
            if pl.t is not None:
                pl.attach(p)
                need = K.klib().vattn_attn_workspace_bytes(C.byref(p))
                w = torch.empty(need // 4 + 1, dtype=torch.float32, device=DEV)
                p.workspace = w.data_ptr()
                keep += [pl, w]
                tag = "  [work list: %d pieces, %d split blocks%s]" % (pl.n_items, pl.n_blocks, ", %d persistent workgroups" % pl.n_wg if pl.n_wg else ", one workgroup per piece")
        ms = time_ms(p, 1, 3 if n > 10000 else 10)
        fl = 4.0 * Hq * 128 * (n * c + n * (n + 1) / 2)
        print("  %-26s n=%6d c=%6d Hq=%2d Hkv=%d : %9.3f ms  %8.1f TFLOP/s  (%.1f%% of 2500)%s" % (name, n, c, Hq, Hkv, ms, fl / ms / 1e9, fl / ms / 1e9 / 25, tag))
        del keep


def decode(variant):
    print("== decode (Sq=1, append + split-KV + combine), %s, D=128 ==" % ("bf16" if DTYPE == torch.bfloat16 else "fp16"))
    for name, Hq, Hkv, B, ctx, slots in [("yi6b B16@32k", 32, 4, 16, 32768, 16), ("yi6b B1@32k", 32, 4, 1, 32768, 4), ("yi6b B4@32k", 32, 4, 4, 32768, 4),
                                         ("yi6b B2@32k", 32, 4, 2, 32768, 4), ("yi6b B8@32k", 32, 4, 8, 32768, 8),
                                         ("yi6b B1@8k", 32, 4, 1, 8192, 4), ("yi6b B1@2k", 32, 4, 1, 2048, 4), ("yi6b B16@2k", 32, 4, 16, 2048, 16),
                                         ("llama8b B64@8k", 32, 8, 64, 8192, 64), ("llama8b B256@2k", 32, 8, 256, 2048, 256),
                                         ("llama70b/tp8 B64@32k", 8, 1, 64, 32768, 64), ("yi34b/tp2 B8@128k", 28, 4, 8, 131072, 8),
                                         ("yi34b/tp2 B1@128k", 28, 4, 1, 131072, 2), ("yi34b/tp4 B1@128k", 14, 2, 1, 131072, 2),
                                         ("mqa G32 B16@16k", 32, 1, 16, 16384, 16), ("gqa G32x4 B8@16k", 128, 4, 8, 16384, 8),
                                         ("mqa G64 B16@8k", 64, 1, 16, 8192, 16)]:
        if ONLY and not any(o.strip() in name for o in ONLY.split(",")):
            continue
        torch.manual_seed(0)
        q = torch.randn(B, 1, Hq, 128, device=DEV, dtype=DTYPE)
        if MEGA > 1:      # megacache layout [slots, ctx, L, Hkv, D]: one layer's view has a row stride of L x Hkv x D elements
            kc = torch.randn(slots, ctx, MEGA, Hkv, 128, device=DEV, dtype=DTYPE)[:, :, MEGA // 2]
            vc = torch.randn(slots, ctx, MEGA, Hkv, 128, device=DEV, dtype=DTYPE)[:, :, MEGA // 2]
        else:
            kc = torch.randn(slots, ctx, Hkv, 128, device=DEV, dtype=DTYPE)
            vc = torch.randn(slots, ctx, Hkv, 128, device=DEV, dtype=DTYPE)
        kn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=DTYPE)
        vn = torch.randn(B, 1, Hkv, 128, device=DEV, dtype=DTYPE)
        cl = torch.full((B,), ctx - 1, dtype=torch.int32, device=DEV)
        idx = torch.arange(B, dtype=torch.int32, device=DEV) % slots
        for splits in SPLITS:
            p, keep = params(q, kc, vc, cl, idx, kn, vn, splits=splits, variant=variant)
            if ROTATE:
                # launches of one shape over R different caches in turn, R x bytes >= 1.5 GB: nothing is served from the 256 MiB
                # Infinity Cache (a small launch repeated on ONE cache is: B1 @ 128 k reads 52 us that way and 63 us in a real model,
                # whose 60 layers each own their K/V)
                by1 = B * 2.0 * ctx * Hkv * 128 * 2
                R = max(2, int(1.5e9 // by1) + 1)
                ps = [(p, keep)]
                for _ in range(R - 1):
                    kc2, vc2 = torch.randn_like(kc), torch.randn_like(vc)
                    ps.append(params(q, kc2, vc2, cl, idx, kn, vn, splits=splits, variant=variant))
                lib = K.klib_for(p.variant)
                st = torch.cuda.current_stream().cuda_stream
                for pp, _k in ps:
                    lib.vattn_flash_attn_with_kvcache(C.byref(pp), st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = max(2, 40 // R + 1)
                e0.record()
                for _ in range(iters):
                    for pp, _k in ps:
                        lib.vattn_flash_attn_with_kvcache(C.byref(pp), st)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / (iters * R)
                del ps
            else:
                ms = time_ms(p, 3, 20)
            by = B * 2.0 * ctx * Hkv * 128 * 2 + B * Hq * 128 * 2 * 2
            print("  %-22s B=%3d ctx=%6d Hq=%2d Hkv=%d splits=%d : %8.4f ms  %7.1f GB/s  (%.1f%% of 8000, %.1f%% of 6290)" % (
                name, B, ctx, Hq, Hkv, splits, ms, by / ms / 1e6, by / ms / 1e6 / 80, by / ms / 1e6 / 62.9))
        del keep, kc, vc


ONLY = None
ROTATE = False
WORKLIST = False
DRAWN = False
PERSIST = True       # work lists walked by persistent workgroups (prefill64p_kernel); --per-piece: one workgroup per piece (prefill64_kernel)
WL_TILES = 0
MEGA = 1
DTYPE = torch.float16
SPLITS = (0,)
PF_SPLITS = 0

if __name__ == "__main__":
    variant = 0
    if "--splits" in sys.argv:
        SPLITS = tuple(int(x) for x in sys.argv[sys.argv.index("--splits") + 1].split(","))
    WORKLIST = "--worklist" in sys.argv
    PERSIST = "--per-piece" not in sys.argv
    DRAWN = "--drawn" in sys.argv        # persistent workgroups draw their pieces (default: host-assigned queues)
    if os.environ.get("KBENCH_PERSIST_MAX_BLOCKS"):      # A/B: let the big balanced grids take a persistent list too
        from vattention_amd import flash_attn as _FA
        _FA.PERSISTENT_MAX_BLOCKS = int(os.environ["KBENCH_PERSIST_MAX_BLOCKS"])
    ROTATE = "--rotate" in sys.argv      # decode: rotate over enough caches that the Infinity Cache cannot serve repeated launches
    WL_TILES = int(sys.argv[sys.argv.index("--wl-tiles") + 1]) if "--wl-tiles" in sys.argv else 0
    if "--mega" in sys.argv:      # decode only: K/V as one layer's view of a megacache tensor with this many layers
        MEGA = int(sys.argv[sys.argv.index("--mega") + 1])
    if "--bf16" in sys.argv:
        DTYPE = torch.bfloat16
    if "--pf-splits" in sys.argv:
        PF_SPLITS = int(sys.argv[sys.argv.index("--pf-splits") + 1])
    if "--only" in sys.argv:
        ONLY = sys.argv[sys.argv.index("--only") + 1]
    if "--variant" in sys.argv:
        variant = int(sys.argv[sys.argv.index("--variant") + 1])
    VARIANTS = [variant] if "--variant" in sys.argv else [0, 8, 12]
    if "--variants" in sys.argv:
        VARIANTS = [int(x) for x in sys.argv[sys.argv.index("--variants") + 1].split(",")]
    what = [a for a in sys.argv[1:] if a in ("prefill", "decode")] or ["prefill", "decode"]
    torch.zeros(1, device=DEV)
    if "prefill" in what:
        for v in VARIANTS:
            print("-- prefill variant %d (order %s, tiling %s) --" % (v, ["XCD-grouped (default)", "block-major per head", "heaviest-first across heads", "XCD-grouped"][(v >> 5) & 3] + (", prefill64 build %d" % ((v >> 8) & 15) if (v >> 8) & 15 else ""), {0: "default plan", 1: "8 waves x 32 rows", 2: "4 waves x 64 rows", 3: "8 waves x 32 rows, LDS-DMA ring, in-wave software pipeline (prefill32)", 4: "4 waves x 32 rows", 6: "8 waves, hand-interleaved MFMA/VALU groups", 7: "4 waves x 64 rows, LDS-DMA ring, in-wave software pipeline (prefill64)"}[(v >> 1) & 7]))
            prefill(v)
    if "decode" in what:
        dvs = [variant]
        if "--dvariants" in sys.argv:
            dvs = [int(x) for x in sys.argv[sys.argv.index("--dvariants") + 1].split(",")]
        for v in dvs:
            print("-- decode variant %d (%s%s) --" % (v, "512-thread workgroups" if v & 65536 else "1024-thread workgroups" if v & 131072 else "256 threads, two K/V register sets per wave" if v & 262144 else "256-thread workgroups (default)",
                                                   ", grid heuristics of rounds 1-3" if v & (1 << 19) else ", device-planned stream, in-launch merge (lab)" if v & (1 << 20) else ", device-planned stream"))
            decode(v)
