"""GPU: BASELINE.json configs[2]'s layout AS WRITTEN — Llama-3-8B (32 layers x 8 kv heads x 128), 64 KiB pages (32 tokens per page,
/root/reference/scripts/benchmark_e2e_dynamic_trace.py:33,58; vattention/utils.h:83-86), one K and one V tensor PER LAYER (no
megacache), max_batch_size 256, 32 k context: 64 virtual tensors of 16 GiB, 1 TiB of address space — on hardware, at the largest pool
whose handle creation stays under ten seconds.

Why not the whole 0.9 x 288 GB: `hipMemCreate` is O(live handles) on ROCm 7.2 (profiles/r01_vmm_scale_probe.txt: 9 us at 5 k handles,
382 us at 50 k) and `hipMemMap` rejects offsets into a larger handle (profiles/r01_vmm_offset_probe.txt), so one 64 KiB page = one
handle and a full pool would be 4 M handles; the bench legs run this model on the megacache layout with 8 MiB pages instead
(vattention_amd/policy.py).  This test shows the written layout WORKING at the real shape — page arithmetic of SURVEY §A.3, on-demand
mapping of 64-page groups, decode over all 8 kv heads of every mapped sequence against the oracle, appended rows bit-exact — not only
at L = 3."""
import time

import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_configs2_written_layout_64kib_pages_full_depth():
    from vattention_amd import vattention as va
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    L, Hq, Hkv, D, B, ctx, page = 32, 32, 8, 128, 256, 32768, 64 << 10
    torch.zeros(1, device=DEV)
    va.enable_layered_async(False)
    tensors = va.init_kvcache(L, Hkv, D, B, ctx, 0, torch.float16, page, False)
    try:
        lay = va.layout()
        # 32 tokens per page, 64 MiB of virtual space per request and tensor, 1024 pages per request
        assert len(tensors) == 2 * L and tuple(tensors[0].shape) == (B, ctx, Hkv, D)
        assert lay["tokens_per_page"] == 32 and lay["virt_bytes_per_req"] == 64 << 20 and lay["max_pages_per_req"] == 1024
        assert lay["virt_bytes_total"] == 16 << 30                       # x 64 tensors = 1 TiB of address space
        groups = 640                                                      # 640 page-groups x 64 pages = 40 960 handles of 64 KiB = 2.5 GiB
        t0 = time.perf_counter()
        npages = va.reserve_physical_pages(groups * 2 * L * page)
        assert npages == groups * 2 * L
        # ragged batch: 20 sequences, 100 .. 1900 tokens (together ~19 k tokens = ~600 groups), in scattered slots
        g = torch.Generator().manual_seed(11)
        lens = [int(x) for x in torch.randint(100, 1900, (20,), generator=g)]
        slots = [int(x) for x in torch.randperm(B, generator=g)[:20]]
        seq_lens = [0] * B
        for s, n in zip(slots, lens):
            seq_lens[s] = n + 1                                           # this step appends one token
        va.step_async(seq_lens)
        va.wait()
        st = va.stats()
        dt = time.perf_counter() - t0
        need = sum((n + 1 + 31) // 32 for n in lens)
        assert st["map_calls"] >= need * 2 * L and st["handles_created"] >= need * 2 * L
        assert dt < 30.0, "creating / mapping %d handles took %.1f s" % (st["handles_created"], dt)
        print("configs[2] layout: %d handles of 64 KiB created, %d map calls, %.2f s (create %.2f s)" % (st["handles_created"], st["map_calls"], dt, st["create_ns"] / 1e9))
        cl = torch.tensor(lens, dtype=torch.int32)
        idx = torch.tensor(slots, dtype=torch.int32)
        ml = max(lens) + 1
        for l in (0, 13, 31):                                             # first, a middle and the last layer's tensors
            k_l, v_l = tensors[l], tensors[L + l]
            torch.manual_seed(100 + l)
            host_k = torch.zeros(B, ml, Hkv, D, dtype=torch.float16)
            host_v = torch.zeros(B, ml, Hkv, D, dtype=torch.float16)
            for s, n in zip(slots, lens):                                 # fill exactly the mapped prefix of every used slot
                host_k[s, :n] = torch.randn(n, Hkv, D).half()
                host_v[s, :n] = torch.randn(n, Hkv, D).half()
                k_l[s, :n].copy_(host_k[s, :n].to(DEV))
                v_l[s, :n].copy_(host_v[s, :n].to(DEV))
            q = torch.randn(20, 1, Hq, D).half()
            kn, vn = torch.randn(20, 1, Hkv, D).half(), torch.randn(20, 1, Hkv, D).half()
            out = flash_attn_with_kvcache(q.to(DEV), k_l[:, :ml], v_l[:, :ml], kn.to(DEV), vn.to(DEV), cache_seqlens=cl.to(DEV),
                                          cache_batch_idx=idx.to(DEV), causal=True)
            torch.cuda.synchronize()
            ref = flash_attn_with_kvcache_ref(q, host_k, host_v, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
            err = (out.double().cpu() - ref).abs()
            assert bool((err <= 2e-3 + 2e-3 * ref.abs()).all()), "layer %d: max err %.3e" % (l, err.max().item())
            for i, (s, n) in enumerate(zip(slots, lens)):                 # the appended row landed in the slot's next row, bit-exact
                assert torch.equal(k_l[s, n].cpu(), kn[i, 0]) and torch.equal(v_l[s, n].cpu(), vn[i, 0])
        # grow every sequence across a page boundary: the next step maps exactly the groups that are missing
        m0 = va.stats()["map_calls"]
        for s, n in zip(slots, lens):
            seq_lens[s] = n + 40
        va.step_async(seq_lens)
        va.wait()
        grown = sum((n + 40 + 31) // 32 - (n + 1 + 31) // 32 for n in lens)
        assert va.stats()["map_calls"] - m0 >= grown * 2 * L
    finally:
        va.cleanup()
