"""GPU: the megacache layout under the attention kernels (row stride L*kvh*D views k[:, :, l]), prefix sharing
through map_common_pages (two slots attending over ONE physical prefix), layer-ordered page mapping, the VMM self-check."""
import pytest
import torch

from oracle.attn import cache_flat_ref, flash_attn_with_kvcache_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _close(got, ref64, what, tol=2e-3):
    err = (got.double().cpu() - ref64).abs()
    bound = tol + tol * ref64.abs()
    assert bool((err <= bound).all()), "%s: max err %.3e" % (what, err.max().item())


def test_attention_over_megacache_views_matches_oracle():
    """Kernel level: K/V as [B, ctx, L, kvh, D] tensors, per-layer views k[:, :, l] (vATTN_cache_engine.py:58-68) — prefill form
    (cache_flat + causal chunk), decode form with fused append through cache_batch_idx, on every layer's view."""
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(7)
    B, ctx, L, Hq, Hkv = 3, 1536, 4, 8, 2
    kmega = torch.randn(B, ctx, L, Hkv, D).half()
    vmega = torch.randn(B, ctx, L, Hkv, D).half()
    kg, vg = kmega.to(DEV), vmega.to(DEV)
    for l in range(L):
        k_l, v_l = kg[:, :, l], vg[:, :, l]                    # strided views: row stride L*Hkv*D
        assert k_l.stride(1) == L * Hkv * D
        kc, vc = kmega[:, :, l].clone(), vmega[:, :, l].clone()  # contiguous host copies for the oracle
        # prefill chunk: n new tokens after c cached ones in slot 1
        n, c, slot = 300 + 17 * l, 700, 1
        q = torch.randn(1, n, Hq, D).half()
        kn, vn = torch.randn(n, Hkv, D).half(), torch.randn(n, Hkv, D).half()
        cache_flat_ref(kn, vn, kc[slot][c:], vc[slot][c:])
        cache_flat(kn.to(DEV), vn.to(DEV), k_l[slot][c:], v_l[slot][c:], "auto")
        cl = torch.tensor([c + n], dtype=torch.int32)
        out = flash_attn_with_kvcache(q.to(DEV), k_l[slot].unsqueeze(0), v_l[slot].unsqueeze(0), cache_seqlens=cl.to(DEV), causal=True)
        ref = flash_attn_with_kvcache_ref(q, kc[slot:slot + 1], vc[slot:slot + 1], cache_seqlens=cl, causal=True)
        _close(out, ref, "megacache prefill layer %d" % l)
        # decode with fused append, ragged lengths, slot indirection, [:, :max_len] view of the view
        lens = torch.tensor([c + n, 1200, 31], dtype=torch.int32)
        idx = torch.tensor([1, 2, 0], dtype=torch.int32)
        qd = torch.randn(3, 1, Hq, D).half()
        kd, vd = torch.randn(3, 1, Hkv, D).half(), torch.randn(3, 1, Hkv, D).half()
        ml = int(lens.max()) + 1
        outd = flash_attn_with_kvcache(qd.to(DEV), k_l[:, :ml], v_l[:, :ml], kd.to(DEV), vd.to(DEV), cache_seqlens=lens.to(DEV),
                                       cache_batch_idx=idx.to(DEV), causal=True)
        refd = flash_attn_with_kvcache_ref(qd, kc[:, :ml], vc[:, :ml], kd, vd, cache_seqlens=lens, cache_batch_idx=idx, causal=True)
        _close(outd, refd, "megacache decode layer %d" % l)
        torch.cuda.synchronize()
        assert torch.equal(k_l.cpu(), kc) and torch.equal(v_l.cpu(), vc)       # appends landed in this layer's rows, bit-exact
    # the other layers' rows were never touched by a layer's append
    for l in range(L):
        assert torch.equal(kg[:, :, l, :, :].cpu()[:, 1400:], kmega[:, 1400:, l])


def _runner_vs_oracle(backend, num_layers, page, prompts, decode_steps, max_len=4096, pool_groups=64, max_batch=4, chunk=None, pass_layer_id=True):
    """Drives engine + wrapper + page manager (HotPathRunner) and checks EVERY layer's output of every iteration with the oracle
    (per-layer activations differ: layer l sees q/k/v scaled by (1 + l/8))."""
    from vattention_amd import vattention
    from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig, Sequence, SequenceMetadata
    Hq, Hkv = 8, 2
    model = ModelConfig(name="tiny", num_layers=num_layers, num_q_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype=torch.float16,
                        max_model_len=max_len, attention_backend=backend)
    group = 2 * page if "megacache" in backend else 2 * num_layers * page
    r = HotPathRunner(model, ParallelConfig(1, 1), CacheConfig(page_size=page, max_batch_size=max_batch, memory_for_gpu=pool_groups * group), seed=11)
    r.sample_kv_util = False
    host = {}
    try:
        def step(mds):
            T = sum(md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1 for md in mds)
            q, k, v = r._qkv(T)
            outs = []
            with torch.cuda.stream(r.stream):
                r.engine.step(mds)
                r.wrapper.begin_forward(mds)
                for l in range(num_layers):
                    s = 1.0 + l / 8.0
                    outs.append(r.wrapper.forward((q * s).half(), (k * s).half(), (v * s).half(), r.engine.gpu_cache[l], r.scale,
                                                  l if pass_layer_id else None))
                r.wrapper.end_forward()
            torch.cuda.synchronize()
            qh, kh, vh = q.float().cpu(), k.float().cpu(), v.float().cpu()
            tok = 0
            for md in mds:
                n = md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len) if md.is_prompt else 1
                for l in range(num_layers):
                    s = 1.0 + l / 8.0
                    kk = (kh[tok:tok + n] * s).half().view(n, Hkv, D)
                    vv = (vh[tok:tok + n] * s).half().view(n, Hkv, D)
                    pk, pv = host.get((md.seq.seq_id, l), (kk[:0], vv[:0]))
                    host[(md.seq.seq_id, l)] = (torch.cat([pk, kk]), torch.cat([pv, vv]))
                    fk, fv = host[(md.seq.seq_id, l)]
                    ref = flash_attn_with_kvcache_ref((qh[tok:tok + n] * s).half().view(1, n, Hq, D), fk.unsqueeze(0).clone(), fv.unsqueeze(0).clone(),
                                                      cache_seqlens=fk.shape[0], causal=True, softmax_scale=D ** -0.5)
                    _close(outs[l][tok:tok + n].view(1, n, Hq, D), ref, "seq %d layer %d ctx %d" % (md.seq.seq_id, l, fk.shape[0]), tol=4e-3)
                tok += n
            for md in mds:
                if md.is_prompt:
                    md.seq.prompt_processed += md.seq.get_next_prompt_chunk_len(md.prompt_chunk_len)
                    if md.seq.prompt_done:
                        md.seq.output_len += 1
                else:
                    md.seq.output_len += 1
            with torch.cuda.stream(r.stream):
                r.engine.on_step_completion(mds)

        seqs = [Sequence(i, p, p + decode_steps + 1) for i, p in enumerate(prompts)]
        for s in seqs:
            while not s.prompt_done:
                step([SequenceMetadata(s, chunk or s.prompt_len, True)])
        for _ in range(decode_steps):
            step([SequenceMetadata(s, 0, False) for s in seqs if not s.is_finished()])
        return vattention.stats()
    finally:
        r.close()


def test_megacache_backend_end_to_end_matches_oracle():
    """fa_vattn_megacache through the cache engine: one 2 MiB page covers all layers (128 tokens per page here), the wrapper gets
    k[:, :, l] views; every layer of every iteration vs the oracle."""
    st = _runner_vs_oracle("fa_vattn_megacache", num_layers=4, page=2 << 20, prompts=[700, 1300], decode_steps=3, chunk=512)
    assert st["map_calls"] > 0 and st["layered_batches"] == 0          # megacache: nothing to order by layer


def test_layer_ordered_mapping_end_to_end_matches_oracle():
    """fa_vattn with 6 layers and 64 KiB pages: a 3000-token prompt needs 24 page-groups, mapped layer-ordered (layers 0-1 before
    step_async returns, 2-5 on the mapper thread while the first layers run, wait_layer gating each layer)."""
    st = _runner_vs_oracle("fa_vattn", num_layers=6, page=64 << 10, prompts=[3000, 900], decode_steps=3, pool_groups=80)
    assert st["layered_batches"] >= 2
    assert st["async_ns"] > 0


def test_layer_ordered_mapping_with_models_that_pass_no_layer_id():
    """The reference's yi / mistral / qwen / falcon / internlm models call wrapper.forward(...) WITHOUT layer_id (only llama.py:179-186
    passes it).  Under the default flags (layer-ordered mapping on) the wrapper must then wait for EVERY layer before the first
    kernel of the iteration: layers >= sync_layers of a new prompt would otherwise run over pages the mapper thread has not mapped
    yet (a GPU memory fault).  Many layers and many small pages so that the mapper is still busy when forward() is entered."""
    from vattention_amd import vattention
    st = _runner_vs_oracle("fa_vattn", num_layers=12, page=64 << 10, prompts=[2900, 1700], decode_steps=2, pool_groups=100,
                           pass_layer_id=False)      # (12 layers x 2 tensors x 46 + 27 pages of 64 KiB: the mapper is busy for milliseconds)
    assert st["layered_batches"] >= 2
    assert vattention._layered_pending is False


def test_two_slots_attend_over_one_shared_prefix():
    """map_common_pages(n) aliases the same physical page-groups under the first pages of EVERY slot (vattention.cu:325-373): a
    prefix written once through slot 0 is the prefix of slot 2 as well.  Both slots then append different suffixes (exclusive
    pages) and attend over prefix + suffix; compared with the oracle.  Refcounted release at the end."""
    from vattention_amd import vattention
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.zeros(1, device=DEV)
    L, Hq, Hkv, B, ctx, page = 2, 8, 2, 3, 4096, 64 << 10       # 128 tokens per page
    ts = vattention.init_kvcache(L, Hkv, D, B, ctx, 0, torch.float16, page, False)
    try:
        vattention.reserve_physical_pages(40 * 2 * L * page)
        P = 256                                                  # two whole pages of shared prefix
        vattention.map_common_pages(P)
        st = vattention.state()
        assert st["mapped"] == [2] * B
        torch.manual_seed(5)
        pk = [torch.randn(P, Hkv, D).half() for _ in range(L)]
        pv = [torch.randn(P, Hkv, D).half() for _ in range(L)]
        for l in range(L):                                       # written ONCE, through slot 0's addresses
            cache_flat(pk[l].to(DEV), pv[l].to(DEV), ts[l][0], ts[L + l][0], "auto")
        torch.cuda.synchronize()
        for l in range(L):                                       # ... and visible under every slot
            for r in range(B):
                assert torch.equal(ts[l][r, :P].cpu(), pk[l]) and torch.equal(ts[L + l][r, :P].cpu(), pv[l])
        free0 = vattention.num_free_kvblocks()
        assert free0 == 40 - 2 + 2                               # the shared groups count once
        lens = [0] * B
        sfx = {0: 200, 2: 333}
        for slot, n in sfx.items():
            lens[slot] = P + n
        vattention.step_async(lens)                               # maps the exclusive pages behind the shared prefix
        for slot, n in sfx.items():
            for l in range(L):
                q = torch.randn(1, n, Hq, D).half()
                k, v = torch.randn(n, Hkv, D).half(), torch.randn(n, Hkv, D).half()
                cache_flat(k.to(DEV), v.to(DEV), ts[l][slot][P:], ts[L + l][slot][P:], "auto")
                cl = torch.tensor([P + n], dtype=torch.int32, device=DEV)
                out = flash_attn_with_kvcache(q.to(DEV), ts[l][slot].unsqueeze(0), ts[L + l][slot].unsqueeze(0), cache_seqlens=cl, causal=True)
                kf = torch.cat([pk[l], k]).unsqueeze(0)
                vf = torch.cat([pv[l], v]).unsqueeze(0)
                ref = flash_attn_with_kvcache_ref(q, kf, vf, cache_seqlens=P + n, causal=True)
                _close(out, ref, "slot %d layer %d over the shared prefix" % (slot, l))
        # the suffix writes went to exclusive pages: the prefix is untouched under the third slot
        torch.cuda.synchronize()
        assert torch.equal(ts[0][1, :P].cpu(), pk[0])
        # everything released: every page id back in the pool exactly once
        vattention.step([0] * B, True)
        st = vattention.state()
        assert st["mapped"] == [0] * B and sorted(st["pool_ids"]) == list(range(st["pool"])) and st["pool"] == 40 * 2 * L
    finally:
        vattention.cleanup()


def test_vmm_selfcheck_passes_and_reports_what_it_saw():
    import ctypes as C
    from vattention_amd import _lib
    torch.zeros(1, device=DEV)
    d = (C.c_uint32 * 3)()
    rc = _lib.lib().vattn_vmm_selfcheck(0, d)
    print("vmm selfcheck rc=%d before=%08x no_flush=%08x after_flush=%08x" % (rc, d[0], d[1], d[2]))
    assert rc == 0 and d[0] == 0xA0A0A0A0 and d[2] == 0xB0B0B0B0
