"""rotary position embedding fused into the attention / cache-append launches, against the CPU restatement of the
reference's stand-alone kernel (oracle/attn.py rotary_embedding_ref <- sarathi-lean/csrc/pos_encoding_kernels.cu:9-77, applied
before the wrapper in models/yi.py:172-173).  The rotated K rows that land in the cache must be BIT-IDENTICAL to the oracle's;
attention outputs within the fp tolerance of tests/test_gpu_attention.py."""
import pytest
import torch

from oracle.attn import cache_flat_ref, flash_attn_with_kvcache_ref, make_cos_sin_cache, rotary_embedding_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 128


def _close(got, ref64, what, tol=2e-3):
    err = (got.double().cpu() - ref64).abs()
    assert bool((err <= tol + tol * ref64.abs()).all()), "%s: max err %.3e" % (what, err.max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_standalone_rotary_kernel_is_bit_exact(dtype):
    from vattention_amd.cache_ops import rotary_embedding
    torch.manual_seed(3)
    T, Hq, Hkv = 300, 8, 2
    cs = make_cos_sin_cache(D, 4096, dtype=dtype)
    for neox in (True, False):
        q, k = torch.randn(T, Hq * D).to(dtype), torch.randn(T, Hkv * D).to(dtype)
        pos = torch.randint(0, 4096, (T,), dtype=torch.int64)
        qg, kg = q.to(DEV), k.to(DEV)
        rotary_embedding(pos.to(DEV), qg, kg, D, cs.to(DEV), neox)
        rotary_embedding_ref(pos, q, k, D, cs, neox)
        torch.cuda.synchronize()
        assert torch.equal(qg.cpu(), q) and torch.equal(kg.cpu(), k), "neox=%s" % neox


@pytest.mark.parametrize("variant", [0, 14, 782], ids=["w8", "dma64", "dma64_xor_image"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_fused_rope_prefill_chunks_and_decode(dtype, variant):
    """Chunked prefill (cache_flat_rope + in-kernel q rotation) then decode steps (q and the appended k rotated in-kernel), slot
    indirection; oracle = rotary kernel on (q, k) at the tokens' positions, then append + attention."""
    from vattention_amd.cache_ops import cache_flat_rope
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(11)
    Hq, Hkv, ctx, slots = 8, 2, 1024, 3
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    cs = make_cos_sin_cache(D, ctx, dtype=dtype)
    csg = cs.to(DEV)
    kc, vc = torch.zeros(slots, ctx, Hkv, D, dtype=dtype), torch.zeros(slots, ctx, Hkv, D, dtype=dtype)
    kg, vg = kc.to(DEV), vc.to(DEV)
    slot, done = 2, 0
    for n in (200, 333, 64):                                   # three chunks of one prompt
        q, k, v = torch.randn(n, Hq * D).to(dtype), torch.randn(n, Hkv * D).to(dtype), torch.randn(n, Hkv * D).to(dtype)
        # GPU: un-rotated q / k go in
        cache_flat_rope(k.to(DEV).view(n, Hkv, D), v.to(DEV).view(n, Hkv, D), kg[slot][done:], vg[slot][done:], csg, done)
        cl = torch.tensor([done + n], dtype=torch.int32)
        out = flash_attn_with_kvcache(q.to(DEV).view(1, n, Hq, D), kg[slot].unsqueeze(0), vg[slot].unsqueeze(0), cache_seqlens=cl.to(DEV),
                                      causal=True, _rotary_cos_sin=csg, _variant=variant)
        # oracle: rotate first (the model does), then the wrapper's dataflow
        qr, kr = q.clone(), k.clone()
        rotary_embedding_ref(torch.arange(done, done + n), qr, kr, D, cs)
        cache_flat_ref(kr.view(n, Hkv, D), v.view(n, Hkv, D), kc[slot][done:], vc[slot][done:])
        ref = flash_attn_with_kvcache_ref(qr.view(1, n, Hq, D), kc[slot:slot + 1], vc[slot:slot + 1], cache_seqlens=cl, causal=True)
        torch.cuda.synchronize()
        assert torch.equal(kg.cpu(), kc) and torch.equal(vg.cpu(), vc), "rotated K rows differ from the oracle's (chunk at %d)" % done
        _close(out, ref, "prefill chunk at %d" % done, tol)
        done += n
    # decode: two sequences (slot 2 continues, slot 0 starts from a short prompt written un-fused), ragged lengths
    kc[0, :37] = torch.randn(37, Hkv, D).to(dtype)
    vc[0, :37] = torch.randn(37, Hkv, D).to(dtype)
    kg[0, :37], vg[0, :37] = kc[0, :37].to(DEV), vc[0, :37].to(DEV)
    lens = [done, 37]
    idx = torch.tensor([2, 0], dtype=torch.int32)
    for step in range(3):
        q, k, v = torch.randn(2, 1, Hq, D).to(dtype), torch.randn(2, 1, Hkv, D).to(dtype), torch.randn(2, 1, Hkv, D).to(dtype)
        cl = torch.tensor(lens, dtype=torch.int32)
        ml = max(lens) + 1
        out = flash_attn_with_kvcache(q.to(DEV), kg[:, :ml], vg[:, :ml], k.to(DEV), v.to(DEV), cache_seqlens=cl.to(DEV),
                                      cache_batch_idx=idx.to(DEV), causal=True, _rotary_cos_sin=csg)
        qr, kr = q.clone().view(2, Hq * D), k.clone().view(2, Hkv * D)
        rotary_embedding_ref(torch.tensor(lens), qr, kr, D, cs)
        ref = flash_attn_with_kvcache_ref(qr.view(2, 1, Hq, D), kc[:, :ml], vc[:, :ml], kr.view(2, 1, Hkv, D), v, cache_seqlens=cl,
                                          cache_batch_idx=idx, causal=True)
        torch.cuda.synchronize()
        assert torch.equal(kg.cpu(), kc) and torch.equal(vg.cpu(), vc), "decode step %d: appended rotated K differs" % step
        _close(out, ref, "decode step %d" % step, tol)
        lens = [x + 1 for x in lens]


def test_wrapper_with_fused_rotary_equals_rotate_then_wrapper():
    """The wrapper with set_fused_rotary(table) fed UN-rotated q / k produces the outputs and the cache contents of the reference's
    dataflow (rotary kernel, then the wrapper) — hybrid iteration with a prefill chunk and two decodes."""
    from vattention_amd.attention import get_attention_wrapper, set_attention_backend
    from vattention_amd.cache_ops import rotary_embedding
    from vattention_amd.replay import ModelConfig, ParallelConfig
    from tests.wrapper_schedule import MD, Seq
    torch.manual_seed(5)
    Hq, Hkv, ctx = 8, 2, 512
    dev = torch.device(DEV)
    model = ModelConfig(name="tiny", num_layers=1, num_q_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype=torch.float16, max_model_len=ctx)
    cs = make_cos_sin_cache(D, ctx).to(dev)
    set_attention_backend("fa_vattn")
    w = get_attention_wrapper()
    w.init(model, ParallelConfig(1, 1), 0, dev)
    outs = {}
    for fused in (True, False):
        torch.manual_seed(6)
        kc = torch.zeros(4, ctx, Hkv, D, dtype=torch.float16, device=dev)
        vc = torch.zeros_like(kc)
        a, b, c = Seq(0, 100, 110), Seq(1, 50, 60), Seq(2, 300, 310)
        res = []
        w.set_fused_rotary(cs if fused else None)
        plan = [([MD(a, 100, True)], [1], []), ([MD(b, 50, True)], [3], []),
                ([MD(c, 128, True), MD(a, 0, False), MD(b, 0, False)], [0], [1, 3]),
                ([MD(c, 128, True), MD(a, 0, False), MD(b, 0, False)], [0], [1, 3])]
        for mds, sp, sd_ in plan:
            T = sum(m.seq.get_next_prompt_chunk_len(m.prompt_chunk_len) if m.is_prompt else 1 for m in mds)
            q = torch.randn(T, Hq * D, device=dev).half()
            k = torch.randn(T, Hkv * D, device=dev).half()
            v = torch.randn(T, Hkv * D, device=dev).half()
            pos, tok = [], 0
            for m in mds:
                if m.is_prompt:
                    n = m.seq.get_next_prompt_chunk_len(m.prompt_chunk_len)
                    pos += list(range(m.seq.prompt_processed, m.seq.prompt_processed + n))
                else:
                    pos.append(m.seq.get_len() - 1)
            if not fused:
                rotary_embedding(torch.tensor(pos, dtype=torch.int64, device=dev), q, k, D, cs, True)      # models/yi.py:172-173
            w.begin_forward(mds)
            w.set_batch_idx(torch.tensor(sp + sd_, dtype=torch.int32, device=dev), torch.tensor(sd_, dtype=torch.int32, device=dev))
            res.append(w.forward(q, k, v, (kc, vc), D ** -0.5, 0).float().cpu())
            w.end_forward()
            for m in mds:
                if m.is_prompt:
                    m.seq.prompt_processed += m.seq.get_next_prompt_chunk_len(m.prompt_chunk_len)
                    if m.seq.prompt_done:
                        m.seq.output_len += 1
                else:
                    m.seq.output_len += 1
        torch.cuda.synchronize()
        outs[fused] = (res, kc.cpu(), vc.cpu())
    w.set_fused_rotary(None)
    assert torch.equal(outs[True][1], outs[False][1]) and torch.equal(outs[True][2], outs[False][2])     # caches bit-identical
    for x, y in zip(outs[True][0], outs[False][0]):
        assert torch.equal(x, y)                                                                            # same rotated operands -> same bits
