"""GPU parity: HIP kernels (through the C ABI, via the flash_attn / cache_ops drop-ins) vs the CPU oracle.

Tolerance: fp16/bf16 I/O compared with the float64 oracle at atol = rtol = 2e-3
(fp16) / 1.6e-2 (bf16, 8 mantissa bits), AND the kernel's max error must stay within 2x the error
of the reference-numerics CPU run (fp32 accumulate, P rounded to the I/O dtype) + 1e-5.
"""
import pytest
import torch

from oracle.attn import cache_flat_ref, flash_attn_with_kvcache_ref
from tests.variants import enabled

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tol(dtype):
    return (2e-3, 2e-3) if dtype == torch.float16 else (1.6e-2, 1.6e-2)


def _check(out_gpu, ref64, ref32, dtype, what):
    atol, rtol = _tol(dtype)
    got = out_gpu.double().cpu()
    err = (got - ref64).abs()
    bound = atol + rtol * ref64.abs()
    assert bool((err <= bound).all()), "%s: max err %.3e (allowed %.3e)" % (what, err.max().item(), bound.max().item())
    e_ref = (ref32.double() - ref64).abs().max().item()
    assert err.max().item() <= 2 * e_ref + 1e-5 + (0 if dtype == torch.float16 else 4e-3), \
        "%s: kernel err %.3e vs reference-numerics err %.3e" % (what, err.max().item(), e_ref)


@pytest.mark.parametrize("variant", [0, 1 << 20, 1 << 19, 1, 65536, 131072, 262144, 1 << 24, 1 << 25, 1 << 27, (1 << 27) | (1 << 20)],
                         ids=["stream", "stream_in_launch_merge", "grid_heuristics", "plain_read", "wg512", "wg1024", "two_register_sets",
                              "striped_everywhere", "single_sequence_contiguous", "xcd_consecutive_ranges", "xcd_local_merge"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("B,G,Hkv,lens", [
    (1, 8, 4, [777]),                    # Yi-6B group size
    (3, 4, 2, [1, 31, 1025]),            # Llama-3-8B group size; tiny and tile-boundary contexts
    (2, 7, 1, [5000, 63]),               # Yi-34B TP2 group size 7 (not a power of two)
    (4, 1, 2, [300, 2, 4095, 64]),       # MHA
])
def test_decode_parity(B, G, Hkv, lens, dtype, variant):
    """variant 0 = the product path: the device-planned stream decomposition (merged by a second launch); bit 20 (lab): merged inside the
    launch; bit 19: the grid heuristics of rounds 1-3; bit 24 (lab): STRIPED pieces in every uniform decomposition (the product stripes the
    single-sequence split launch only), bit 25 (lab): contiguous pieces there too; the rest are lab shapes.  num_splits < 0 forces that many workgroups per kv head on the
    stream path (pieces that cross sequence boundaries, ranges that end inside a tile space smaller than the grid, ...)."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(1234)
    Hq, D, ctx, slots = G * Hkv, 128, 6000, 7
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    q = torch.randn(B, 1, Hq, D).to(dtype)
    kn = torch.randn(B, 1, Hkv, D).to(dtype)
    vn = torch.randn(B, 1, Hkv, D).to(dtype)
    idx = torch.tensor([5, 0, 3, 6][:B], dtype=torch.int32)
    cl = torch.tensor(lens, dtype=torch.int32)
    max_len = max(lens) + 1
    kc1, vc1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, kc1[:, :max_len], vc1[:, :max_len], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    kc2, vc2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, kc2[:, :max_len], vc2[:, :max_len], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    kg, vg = kc.to(DEV), vc.to(DEV)
    stream = variant in (0, 1 << 20, 1 << 24, 1 << 25, 1 << 27, (1 << 27) | (1 << 20))
    for splits in (0, 1, 3) + ((-1, -2, -5, -37, -300) if stream else ()):
        kgi, vgi = kg.clone(), vg.clone()
        for rep in range(2 if stream else 1):          # (the in-launch merge's tickets reset themselves: a second call must work too)
            out = flash_attn_with_kvcache(q.to(DEV), kgi[:, :max_len], vgi[:, :max_len], kn.to(DEV), vn.to(DEV),
                                          cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV), causal=True,
                                          num_splits=splits, _variant=variant)
            torch.cuda.synchronize()
            _check(out, ref64, ref32, dtype, "decode splits=%d call %d" % (splits, rep))
        assert torch.equal(kgi.cpu(), kc1) and torch.equal(vgi.cpu(), vc1)     # in-place append, bit-exact, nothing else touched


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("B,G,Hkv,D,lens", [
    (1, 8, 4, 128, [32700]),              # B1 @ 32k: 48 splits, the shape the single-launch merge exists for
    (3, 8, 2, 128, [5000, 17, 900]),      # ragged: some splits of the short sequences are empty
    (2, 40, 1, 128, [3000, 777]),         # G = 40: two workgroups of two 16-head blocks each (the second half empty in the last)
    (2, 32, 2, 128, [4096, 100]),         # G = 32: ONE pass over K/V for all 32 heads
    (2, 71, 1, 64, [2500, 300]),          # Falcon-7B shape (d = 64, G = 71)
    (4, 17, 2, 64, [1000, 1, 64, 333]),
], ids=["b1_32k", "ragged", "g40", "g32", "falcon_g71_d64", "g17_d64"])
def test_decode_single_launch_merge_and_head_block_groups(B, G, Hkv, D, lens, dtype):
    """Round 2 decode forms against the oracle: (a) the split-KV merge inside the decode launch (variant bit 9 forces it, bit 8
    forbids it, default = by grid size), incl. repeated calls on one stream (the group counters reset themselves); (b) two 16-head
    blocks per workgroup for G > 16 (variant bit 7 = one block per workgroup, the round-1 form).  The in-launch merge is opt-in
    (measured slower: agent-scope fences flush the L2); default and bit 8 are the two-launch form."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(B * 131 + G)
    Hq, ctx, slots = G * Hkv, max(lens) + 40, B + 2
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    q = torch.randn(B, 1, Hq, D).to(dtype)
    kn = torch.randn(B, 1, Hkv, D).to(dtype)
    vn = torch.randn(B, 1, Hkv, D).to(dtype)
    idx = torch.randperm(slots)[:B].to(torch.int32)
    cl = torch.tensor(lens, dtype=torch.int32)
    ml = max(lens) + 1
    kc1, vc1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, kc1[:, :ml], vc1[:, :ml], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    kc2, vc2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, kc2[:, :ml], vc2[:, :ml], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    outs = {}
    for variant in enabled(0, 256, 512, 1024, 128, 128 | 512, 128 | 1024):      # bit 9: merge ordered by fences, bit 10: device-scope accesses
        for splits in (0, 5):
            kg, vg = kc.to(DEV), vc.to(DEV)
            for rep in range(3 if variant & (512 | 1024) else 1):
                out = flash_attn_with_kvcache(q.to(DEV), kg[:, :ml], vg[:, :ml], kn.to(DEV), vn.to(DEV), cache_seqlens=cl.to(DEV),
                                              cache_batch_idx=idx.to(DEV), causal=True, num_splits=splits, _variant=variant)
                torch.cuda.synchronize()
                _check(out, ref64, ref32, dtype, "decode variant %d splits %d call %d" % (variant, splits, rep))
            assert torch.equal(kg.cpu(), kc1) and torch.equal(vg.cpu(), vc1)
            outs[(variant, splits)] = out.float().cpu()
    # the merge arithmetic is the same in both forms: same values up to the order of one multiply-add
    for splits in (0, 5):
        if (512, splits) not in outs:      # product-only run (-m "gpu and not lab"): the in-launch merges are lab builds
            continue
        assert (outs[(256, splits)] - outs[(512, splits)]).abs().max().item() <= (1e-3 if dtype == torch.float16 else 8e-3)
        assert torch.equal(outs[(512, splits)], outs[(1024, splits)]), "the two in-launch merge protocols differ only in how the partials travel"


@pytest.mark.parametrize("variant", [0, 1, 8, 4, 16, 12, 14, 6, 782, 526], ids=["w8q1_tr", "w8q1_plain", "w4q1_tr", "w4q2_tr", "w8q1_mfma_rowsum", "w8_interleaved", "w4q2_dma_pipelined", "w8q1_dma_pipelined", "w4q2_dma_xor_image", "w4q2_dma_kpad"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("n,c,Hq,Hkv", [
    (128, 0, 8, 2), (200, 0, 4, 1), (77, 333, 8, 4), (512, 1000, 4, 2), (130, 62, 2, 2), (64, 64, 4, 2),
])
def test_prefill_chunk_parity(n, c, Hq, Hkv, dtype, variant):
    """Wrapper's prefill form (:151-166): cache_flat at offset c, then causal attention with cache_seqlens=[c+n]."""
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(n * 7 + c)
    D, ctx = 128, 1600
    kc = torch.randn(3, ctx, Hkv, D).to(dtype)       # pre-existing cache rows [0, c)
    vc = torch.randn(3, ctx, Hkv, D).to(dtype)
    q = torch.randn(1, n, Hq, D).to(dtype)
    k = torch.randn(n, Hkv, D).to(dtype)
    v = torch.randn(n, Hkv, D).to(dtype)
    slot = 1
    cl = torch.tensor([c + n], dtype=torch.int32)
    kc1, vc1 = kc.clone(), vc.clone()
    cache_flat_ref(k, v, kc1[slot][c:], vc1[slot][c:])
    ref64 = flash_attn_with_kvcache_ref(q, kc1[slot:slot + 1], vc1[slot:slot + 1], cache_seqlens=cl, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc1[slot:slot + 1], vc1[slot:slot + 1], cache_seqlens=cl, causal=True, math="f32")
    kg, vg = kc.to(DEV), vc.to(DEV)
    key_cache = kg[slot].reshape(1, -1, Hkv, D)
    value_cache = vg[slot].reshape(1, -1, Hkv, D)
    cache_flat(k.to(DEV), v.to(DEV), key_cache.squeeze(0)[c:], value_cache.squeeze(0)[c:], "auto")
    out = flash_attn_with_kvcache(q.to(DEV), key_cache, value_cache, cache_seqlens=cl.to(DEV), causal=True, _variant=variant)
    torch.cuda.synchronize()
    assert torch.equal(kg.cpu(), kc1) and torch.equal(vg.cpu(), vc1)           # cache_flat bit-exact
    _check(out, ref64, ref32, dtype, "prefill n=%d c=%d" % (n, c))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("Hq,Hkv,chunks", [
    (8, 1, [(0, 2048)]),                                  # a TP=8 shard's whole prompt: the upper query blocks are cut, the lower are not
    (8, 1, [(0, 1500)]),                                  # ragged last block
    (8, 1, [(3000, 700)]),                                # a chunk on a prefix: equal blocks, all cut
    (8, 2, [(0, 1300), (200, 1), (640, 300)]),            # batched chunks of different lengths incl. a one-token entry (varlen form)
    (4, 4, [(0, 520), (0, 2100)]),                        # MHA, two prompts
], ids=["tp8_2k", "tp8_1500", "chunk700_at_3000", "varlen3", "mha2"])
def test_prefill_work_list_matches_the_oracle(Hq, Hkv, chunks, dtype):
    """Prefill launches driven by a host-planned work list (vattn_prefill_plan: pieces longest first, only long query blocks cut,
    partials merged by combine_blocks_kernel) against the oracle, and against the default launch of the same call."""
    from vattention_amd import flash_attn as FA
    from vattention_amd import kernels as K
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache
    torch.manual_seed(31)
    D, P = 128, len(chunks)
    ctx = max(c + n for c, n in chunks) + 9
    slots = P + 2
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    T = sum(n for _, n in chunks)
    q = torch.randn(T, Hq, D).to(dtype)
    sl = torch.randperm(slots)[:P].to(torch.int32)
    q_lens, k_lens = [n for _, n in chunks], [c + n for c, n in chunks]
    refs64, refs32, tok = [], [], 0
    for i, (c, n) in enumerate(chunks):
        s_ = int(sl[i])
        for math, dst in (({}, refs64), ({"math": "f32"}, refs32)):
            dst.append(flash_attn_with_kvcache_ref(q[tok:tok + n].unsqueeze(0), kc[s_:s_ + 1].clone(), vc[s_:s_ + 1].clone(),
                                                   cache_seqlens=torch.tensor([c + n], dtype=torch.int32), causal=True, **math)[0])
        tok += n
    ref64, ref32 = torch.cat(refs64), torch.cat(refs32)
    kg, vg, qg = kc.to(DEV), vc.to(DEV), q.to(DEV)
    p = K.AttnParams()
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = P, max(q_lens), Hq, Hkv, D, 1
    # (launches this small keep the default plan on their own: pieces of at most 9 tiles are forced, as a long prompt would get)
    plan = FA.prefill_plan(p, q_lens, k_lens, torch.device(DEV), force_tiles=9)
    assert plan.t is not None and plan.n_items > 0 and plan.n_blocks > 0, "no work list"
    natural = FA.prefill_plan(p, q_lens, k_lens, torch.device(DEV))      # the planner's own choice: always a list for ragged batches (valid blocks only)
    if P > 1 and len({(n + 255) // 256 for n in q_lens}) > 1:
        assert natural.t is not None and natural.n_items >= sum((n + 255) // 256 for n in q_lens) * Hq
        per_piece = FA.prefill_plan(p, q_lens, k_lens, torch.device(DEV), persistent=False)      # one workgroup per piece: compact and uncut
        assert per_piece.t is not None and per_piece.n_blocks == 0 and per_piece.n_items == sum((n + 255) // 256 for n in q_lens) * Hq
    outs = []
    for pl in (plan, None) + ((natural,) if natural.t is not None else ()):
        out = torch.full((T, Hq, D), float("nan"), dtype=dtype, device=DEV)
        if P == 1:
            s_ = int(sl[0])
            flash_attn_with_kvcache(qg.unsqueeze(0), kg[s_:s_ + 1], vg[s_:s_ + 1], cache_seqlens=torch.tensor(k_lens, dtype=torch.int32, device=DEV),
                                    causal=True, out=out.unsqueeze(0), _max_seqlen_k=k_lens[0], _pf_plan=pl)
        else:
            starts = torch.tensor([sum(q_lens[:i]) for i in range(P)], dtype=torch.int32, device=DEV)
            flash_attn_varlen_with_kvcache(qg, kg, vg, starts, torch.tensor(q_lens, dtype=torch.int32, device=DEV), max(q_lens),
                                           torch.tensor(k_lens, dtype=torch.int32, device=DEV), sl.to(DEV), causal=True, out=out,
                                           _max_seqlen_k=max(k_lens), _pf_plan=pl)
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any(), "rows left unwritten (%s)" % ("work list" if pl else "default launch")
        _check(out, ref64, ref32, dtype, "prefill, %s" % ("work list: %d pieces, %d split blocks" % (plan.n_items, plan.n_blocks) if pl else "default launch"))
        outs.append(out.float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= (2e-3 if dtype == torch.float16 else 1.6e-2)
    if len(outs) > 2:      # (the default launch of so small a shape may take another tiling: same values up to fp rounding)
        assert (outs[2] - outs[1]).abs().max().item() <= (2e-3 if dtype == torch.float16 else 1.6e-2)


@pytest.mark.parametrize("variant", [0, 65536, 131072, 262144], ids=["wg256", "wg512", "wg1024", "two_register_sets"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("Hq,Hkv,lens", [
    (8, 1, [5000, 31, 900, 2100, 64, 1, 3333, 12000, 700, 450]),          # one TP=8 rank: ragged contexts on ONE kv head
    (32, 8, [9000, 400, 2048, 6000]),                                      # Llama-3-8B heads
    (14, 2, [11000, 100, 100, 100, 100, 100, 5000]),                       # G = 7
    (32, 1, [4000, 333, 2500]),                                            # G = 32: two 16-head blocks per workgroup
], ids=["tp8_rank", "llama8b", "g7", "g32"])
def test_decode_length_balanced_plan(Hq, Hkv, lens, dtype, variant):
    """A ragged decode batch through the length-balanced plan (vattn_decode_plan: `_cache_seqlens_host` on the Python side) — pieces
    of near-equal length instead of the same number of splits for every sequence — against the oracle, fused append bit-exact, and
    against the uniform split of the same call (same values up to the order of the fp32 merge)."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    import vattention_amd.flash_attn as FA
    torch.manual_seed(99)
    B, D, ctx = len(lens), 128, max(lens) + 40
    slots = B + 3
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    q = torch.randn(B, 1, Hq, D).to(dtype)
    kn = torch.randn(B, 1, Hkv, D).to(dtype)
    vn = torch.randn(B, 1, Hkv, D).to(dtype)
    idx = torch.randperm(slots)[:B].to(torch.int32)
    cl = torch.tensor(lens, dtype=torch.int32)
    max_len = max(lens) + 1
    kc1, vc1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, kc1[:, :max_len], vc1[:, :max_len], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    kc2, vc2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, kc2[:, :max_len], vc2[:, :max_len], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    outs = []
    # (a batch this small keeps the uniform split on its own: pieces of 7 / 40 tiles are forced, as a batch of hundreds would get)
    for host, tiles in ((lens, 7), (lens, 40), (None, 0)):
        kg, vg = kc.to(DEV), vc.to(DEV)
        cap = []
        out = flash_attn_with_kvcache(q.to(DEV), kg[:, :max_len], vg[:, :max_len], kn.to(DEV), vn.to(DEV), cache_seqlens=cl.to(DEV),
                                      cache_batch_idx=idx.to(DEV), causal=True, _variant=variant, _cache_seqlens_host=host, _params_out=cap,
                                      _plan_tiles=tiles)
        torch.cuda.synchronize()
        if host is not None:
            assert cap[0].num_split_items >= B + (2 if tiles == 7 else 1), "the ragged batch did not get a plan"
        else:
            assert cap[0].num_split_items == 0
        _check(out, ref64, ref32, dtype, "ragged decode, %s" % ("length-balanced plan, %d tiles per piece" % tiles if host else "uniform split"))
        assert torch.equal(kg.cpu(), kc1) and torch.equal(vg.cpu(), vc1)
        # the same parameter block again (what layers 1..L-1 of an iteration do)
        out2 = torch.empty_like(out)
        FA.relaunch(cap[0], q.to(DEV).data_ptr(), kn.to(DEV).data_ptr(), vn.to(DEV).data_ptr(), out2.data_ptr(), kg.data_ptr(), vg.data_ptr(), torch.device(DEV))
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    for o in outs[:2]:
        assert (o - outs[2]).abs().max().item() <= (2e-3 if dtype == torch.float16 else 1.6e-2)


@pytest.mark.parametrize("Hq,Hkv,cs,cl,bs", [(32, 8, 512, 4096, 8), (8, 1, 1024, 8192, 4), (16, 2, 2048, 4096, 6)],
                         ids=["llama3_8b_tp1", "yi6b_tp4", "yi6b_tp2"])
def test_pod_sweep_shapes_at_the_reference_tolerance(Hq, Hkv, cs, cl, bs):
    """The only numeric assertion in the reference tree for this operator is GPU-vs-GPU: POD's fused launch against FlashAttention at
    torch.allclose(atol = 1e-3) on fp16 N(0,1) inputs (pod_attn/tests/attn_sweep.py:82-97; shapes :8-69: a chunk of `cs` tokens whose
    keys are the first chunks of a `cl`-token prompt, plus `bs` decodes at cl - 1).  The same shapes here, HIP kernels against the
    oracle in the reference kernel's numerics (fp32 accumulate, P rounded to fp16), at the SAME criterion."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(cs + cl)
    D, dtype = 128, torch.float16
    cache_seqlen = 2 * cs if 2 * cs <= cl else cs                      # the sweep's second chunk (chunk_idx = 1)
    q_p = torch.randn(1, cs, Hq, D).to(dtype)
    k_p = torch.randn(1, cache_seqlen, Hkv, D).to(dtype)
    v_p = torch.randn(1, cache_seqlen, Hkv, D).to(dtype)
    q_d = torch.randn(bs, 1, Hq, D).to(dtype)
    k_d = torch.randn(bs, cl, Hkv, D).to(dtype)
    v_d = torch.randn(bs, cl, Hkv, D).to(dtype)
    lens_p = torch.tensor([cache_seqlen], dtype=torch.int32)
    lens_d = torch.tensor([cl - 1] * bs, dtype=torch.int32)
    ref_p = flash_attn_with_kvcache_ref(q_p, k_p, v_p, cache_seqlens=lens_p, causal=True, math="f32")
    ref_d = flash_attn_with_kvcache_ref(q_d, k_d, v_d, cache_seqlens=lens_d, causal=True, math="f32")
    out_p = flash_attn_with_kvcache(q_p.to(DEV), k_p.to(DEV), v_p.to(DEV), cache_seqlens=lens_p.to(DEV), causal=True)
    out_d = flash_attn_with_kvcache(q_d.to(DEV), k_d.to(DEV), v_d.to(DEV), cache_seqlens=lens_d.to(DEV), causal=True)
    torch.cuda.synchronize()
    assert torch.allclose(out_p.cpu().float(), ref_p.float(), atol=1e-3), "prefill output mismatch: %.3e" % (out_p.cpu().float() - ref_p.float()).abs().max().item()
    assert torch.allclose(out_d.cpu().float(), ref_d.float(), atol=1e-3), "decode output mismatch: %.3e" % (out_d.cpu().float() - ref_d.float()).abs().max().item()


@pytest.mark.parametrize("B,lo,hi,Hkv,nwg", [(256, 4000, 29000, 1, 0), (16, 32767, 32767, 4, 0), (40, 1, 3000, 8, -7), (9, 10, 2500, 2, -11), (100, 0, 900, 1, -50)],
                         ids=["ragged256", "uniform16", "spanning", "few", "sparse"])
def test_decode_stream_plan_table_matches_the_model(B, lo, hi, Hkv, nwg):
    """The plan the decode launch derives ON THE DEVICE — the (first record, count) of every sequence, left at the head of the launch's
    workspace for the merge launch — against its CPU restatement (oracle/stream_plan.py), for the product's own workgroup count
    (vattn_attn_plan_describe) and forced ones."""
    import ctypes as C
    import vattention_amd.flash_attn as FA
    from oracle.stream_plan import plan as model
    from vattention_amd import kernels as K
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    g = torch.Generator().manual_seed(B + hi)
    lens = torch.randint(lo, hi + 1, (B,), generator=g).tolist()
    Hq, D = Hkv * 4, 128
    ml = max(lens) + 1
    kc = torch.randn(B, ml, Hkv, D, device=DEV, dtype=torch.float16)
    vc = torch.randn(B, ml, Hkv, D, device=DEV, dtype=torch.float16)
    q = torch.randn(B, 1, Hq, D, device=DEV, dtype=torch.float16)
    kn, vn = torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16), torch.randn(B, 1, Hkv, D, device=DEV, dtype=torch.float16)
    cap = []
    FA._workspaces.clear()
    flash_attn_with_kvcache(q, kc, vc, kn, vn, cache_seqlens=torch.tensor(lens, dtype=torch.int32, device=DEV), causal=True, num_splits=nwg, _params_out=cap)
    torch.cuda.synchronize()
    d = K.describe(cap[0])
    assert d["path"] == 2
    per_head = d["workgroups"] // Hkv
    uniform, pieces, records = model(lens, ml, 1, per_head)
    ws = next(iter(FA._workspaces.values()))
    table = ws[:2 * B].view(torch.int32).cpu().view(B, 2).tolist()
    assert [tuple(r) for r in table] == records, "device plan differs from the model (uniform=%s, %d workgroups per kv head)" % (uniform, per_head)


def test_decode_stream_plan_without_cache_seqlens_or_batch_idx():
    """cache_seqlens = None (every sequence is the whole cache view) and cache_batch_idx = None (identity): the device-side plan takes
    both defaults (flash_attn_interface.py:1168-1254: both arguments are optional)."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(21)
    B, Hq, Hkv, D, ctx = 5, 8, 2, 128, 1500
    q = torch.randn(B, 1, Hq, D).half()
    kc, vc = torch.randn(B, ctx, Hkv, D).half(), torch.randn(B, ctx, Hkv, D).half()
    ref64 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=torch.full((B,), ctx, dtype=torch.int32), causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=torch.full((B,), ctx, dtype=torch.int32), causal=True, math="f32")
    for nwg in (0, -3, -40):
        out = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), causal=True, num_splits=nwg)
        torch.cuda.synchronize()
        _check(out, ref64, ref32, torch.float16, "decode without cache_seqlens, num_splits=%d" % nwg)


def _op_names():
    from tests.test_attn_oracle import _intree_cases
    return [n for n in _intree_cases()[1] if n.startswith("op_")]


@pytest.mark.parametrize("name", _op_names())
def test_kernels_against_the_operator_by_composition_vectors(name):
    """The HIP kernels, directly against the vectors composed from reference text (tests/golden/attn_intree_ref_mha.npz, OP_CASES of
    oracle/gen_golden_attn_intree.py: append by the reference's cache_flat statement, slot by cache_batch_idx, keys cut at cache_seqlens,
    attention by the in-tree ref_mha_bmhk) — no oracle in between.  Tolerances: the fp16 / bf16 output rounding (2e-3 / 1.6e-2), AND the
    reference's own GPU-vs-GPU criterion atol = 1e-3 (pod_attn/tests/attn_sweep.py:82-97) for fp16; appended rows bit-exact."""
    from tests.test_attn_oracle import _op_case
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    z, case, kc, vc, q, kn, vn = _op_case(name)
    _n, B, Sq, Sn, lens, slots, idx, Hq, Hkv, D, causal, dtype = case
    dt = getattr(torch, dtype)
    cl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    bi = torch.tensor(idx, dtype=torch.int32, device=DEV) if idx is not None else None
    ml = max(lens) + Sn
    kg, vg = kc.to(DEV), vc.to(DEV)
    kview, vview = (kg[:, :ml], vg[:, :ml]) if "strided" in name or Sq == 1 else (kg, vg)      # (the decode call site passes the [:, :max_len] view)
    out, lse = flash_attn_with_kvcache(q.to(DEV), kview, vview, kn.to(DEV) if Sn else None, vn.to(DEV) if Sn else None, cache_seqlens=cl,
                                       cache_batch_idx=bi, causal=bool(causal), return_softmax_lse=True)
    torch.cuda.synchronize()
    ref = torch.from_numpy(z[name + "/out"]).double()
    got = out.double().cpu()
    atol, rtol = _tol(dt)
    err = (got - ref).abs()
    assert bool((err <= atol + rtol * ref.abs()).all()), "%s: max err %.3e" % (name, err.max().item())
    if dt == torch.float16:
        assert torch.allclose(got, ref, atol=1e-3, rtol=1e-3), "%s: fails the reference's own atol = 1e-3 criterion: %.3e" % (name, err.max().item())
    dead = torch.from_numpy(z[name + "/masked_rows"])
    if dead.any():
        assert float(got[dead].abs().max()) == 0.0
    live = ~dead[:, None, :].expand(-1, Hq, -1)
    assert torch.allclose(lse.double().cpu()[live], torch.from_numpy(z[name + "/lse"]).double()[live], atol=2e-3, rtol=2e-3)
    # the caches after the in-kernel append = what the reference's cache_flat statement left
    assert torch.equal(kg.cpu().view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)), torch.from_numpy(z[name + "/k_sum"]))
    assert torch.equal(vg.cpu().view(torch.int16).to(torch.int64).sum(dim=(1, 2, 3)), torch.from_numpy(z[name + "/v_sum"]))


@pytest.mark.parametrize("append", [True, False], ids=["append", "no_append"])
@pytest.mark.parametrize("Hq,Hkv,B,lo,hi,nwg", [
    (8, 1, 250, 0, 900, 0),            # four sequences per lane of the planning wave, some EMPTY
    (8, 1, 250, 0, 900, -700),         # ... and far more workgroups than the tile space has tiles
    (8, 1, 100, 0, 900, -50),          # two sequences per lane
    (32, 8, 40, 1, 3000, 0),           # Llama-3-8B heads
    (32, 8, 40, 1, 3000, -7),          # every workgroup's range spans several sequences
    (28, 4, 1, 9000, 9000, -64),       # ONE sequence (forced onto the stream path: the product takes the uniform grid for B = 1): 64 pieces
    (32, 1, 9, 10, 2500, -11),         # G = 32: two 16-head blocks per workgroup
    (8, 2, 256, 0, 70, 0),             # the largest batch the plan prologue takes
    (8, 2, 300, 0, 70, 0),             # ... and one beyond it (the launch falls back to the grid heuristics)
], ids=["b250", "b250_sparse", "b100", "llama8b", "llama8b_spanning", "one_sequence", "g32", "b256", "b300_fallback"])
def test_decode_stream_plan(Hq, Hkv, B, lo, hi, nwg, append):
    """The device-planned stream decomposition (csrc/decode_body.h, decode_stream_kernel) on batches that stress the PLAN: every sequence
    against the oracle (f64 and reference numerics), the appended rows bit-exact, the lab build's in-launch merge against the product's
    two-launch form (the same arithmetic up to the chunking of the sum), repeated calls on one stream (the tickets reset themselves).  No host-side lengths anywhere."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(B * 7 + Hq)
    dtype, D = torch.float16, 128
    g = torch.Generator().manual_seed(B + hi)
    lens = torch.randint(lo, hi + 1, (B,), generator=g).tolist()
    if B >= 100:
        lens[0] = lens[17] = lens[B - 1] = 0            # empty sequences (first, middle, last)
    ctx = max(lens) + 8
    slots = B + 3
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    q = torch.randn(B, 1, Hq, D).to(dtype)
    kn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
    vn = torch.randn(B, 1, Hkv, D).to(dtype) if append else None
    idx = torch.randperm(slots, generator=g)[:B].to(torch.int32)
    cl = torch.tensor(lens, dtype=torch.int32)
    ml = max(lens) + (1 if append else 0)
    kc1, vc1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, kc1[:, :ml], vc1[:, :ml], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    kc2, vc2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, kc2[:, :ml], vc2[:, :ml], kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    dev = lambda t: t.to(DEV) if t is not None else None
    outs = {}
    for variant in enabled(0, 1 << 20, 1 << 27, (1 << 27) | (1 << 20), 1 << 21):      # bit 27 (lab): the ranges of one XCD are consecutive; with bit 20 a sequence inside one XCD is handed over in its L2
        kg, vg = kc.to(DEV), vc.to(DEV)
        for rep in range(2):
            out, lse = flash_attn_with_kvcache(dev(q), kg[:, :ml], vg[:, :ml], dev(kn), dev(vn), cache_seqlens=dev(cl), cache_batch_idx=dev(idx),
                                               causal=True, num_splits=nwg, _variant=variant, return_softmax_lse=True)
            torch.cuda.synchronize()
            _check(out, ref64, ref32, dtype, "stream decode, variant %d, call %d" % (variant, rep))
        assert torch.equal(kg.cpu(), kc1) and torch.equal(vg.cpu(), vc1)
        outs[variant] = (out.cpu(), lse.cpu())
    # (the two merges fold the records in chunks of 16 / 8: the same sum up to the order of a few fp32 multiply-adds)
    for v in enabled(1 << 20, 1 << 27, (1 << 27) | (1 << 20), 1 << 21):      # bit 21 (lab, round 6): a drawn queue of fixed 16-tile pieces
        assert (outs[0][0].float() - outs[v][0].float()).abs().max().item() <= 1e-3 and torch.allclose(outs[0][1], outs[v][1], atol=1e-4, rtol=1e-5), v
    if not append:      # an empty sequence attends to nothing: zeros, LSE = +inf as FlashAttention reports it
        for b, l in enumerate(lens):
            if l == 0:
                assert float(outs[0][0][b].abs().max()) == 0.0 and bool(torch.isinf(outs[0][1][b]).all())
    # natural-log LSE against the oracle's scores
    got = outs[0][1][:, :, 0].double()
    for b in (1, B // 2, B - 2) if B > 3 else (0,):
        Lk = lens[b] + (1 if append else 0)
        if Lk == 0:
            continue
        keys = kc1[idx[b].item(), :Lk].double()                       # [Lk, Hkv, D] (after the oracle's append)
        qq = q[b, 0].double().view(Hkv, Hq // Hkv, D)
        sc = torch.einsum("kgd,lkd->kgl", qq, keys) * D ** -0.5
        want = torch.logsumexp(sc, dim=-1).reshape(-1)
        assert torch.allclose(got[b], want, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("c,Hq,Hkv", [(5, 4, 4), (0, 8, 2), (777, 32, 4), (1599, 8, 1)])
def test_one_token_prefill_chunk(c, Hq, Hkv, dtype):
    """A ONE-token prefill chunk — what the Sarathi scheduler issues when a prompt's remainder is a single token
    (vattention_flashattention_wrapper.py:151-166: cache_flat at offset c, then q [1, 1, Hq, D] with causal=True, cache_seqlens=[c+1],
    no k / v arguments).  seqlen_q == 1 takes the decode form of the launch (causal is moot for one row, flash_api.cpp:1364) without
    the in-kernel append; the token attends over the c cached keys and itself."""
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(1000 + c)
    D, ctx = 128, 1600
    kc = torch.randn(3, ctx, Hkv, D).to(dtype)
    vc = torch.randn(3, ctx, Hkv, D).to(dtype)
    q = torch.randn(1, 1, Hq, D).to(dtype)
    k = torch.randn(1, Hkv, D).to(dtype)
    v = torch.randn(1, Hkv, D).to(dtype)
    slot = 2
    cl = torch.tensor([c + 1], dtype=torch.int32)
    kc1, vc1 = kc.clone(), vc.clone()
    cache_flat_ref(k, v, kc1[slot][c:], vc1[slot][c:])
    ref64 = flash_attn_with_kvcache_ref(q, kc1[slot:slot + 1], vc1[slot:slot + 1], cache_seqlens=cl, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc1[slot:slot + 1], vc1[slot:slot + 1], cache_seqlens=cl, causal=True, math="f32")
    kg, vg = kc.to(DEV), vc.to(DEV)
    key_cache, value_cache = kg[slot].unsqueeze(0), vg[slot].unsqueeze(0)
    cache_flat(k.to(DEV), v.to(DEV), key_cache.squeeze(0)[c:], value_cache.squeeze(0)[c:], "auto")
    out = torch.full((1, 1, Hq, D), float("nan"), dtype=dtype, device=DEV)
    flash_attn_with_kvcache(q.to(DEV), key_cache, value_cache, cache_seqlens=cl.to(DEV), causal=True, out=out, _max_seqlen_k=c + 1)
    torch.cuda.synchronize()
    assert torch.equal(kg.cpu(), kc1) and torch.equal(vg.cpu(), vc1)           # cache_flat bit-exact, nothing else written
    _check(out, ref64, ref32, dtype, "one-token chunk at c=%d" % c)


@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (8, 2), (8, 4), (8, 8), (6, 3), (16, 16), (12, 4)])
def test_prefill_workgroup_orders(Hq, Hkv):
    """The three workgroup->(batch, head, query block) mappings (variant bits 5-6; XCD-grouped default, its fallback for
    kv-head counts that do not divide the 8 XCDs) cover every (b, h, block) exactly once: parity with the oracle on a ragged
    batch with several query blocks, and bit-identical outputs across mappings."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(Hq * 31 + Hkv)
    B, n, D, ctx = 3, 600, 128, 1400
    q = torch.randn(B, n, Hq, D).half()
    kc = torch.randn(4, ctx, Hkv, D).half()
    vc = torch.randn(4, ctx, Hkv, D).half()
    cl = torch.tensor([n + 700, n, n + 129], dtype=torch.int32)
    idx = torch.tensor([2, 0, 3], dtype=torch.int32)
    ref64 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    outs = []
    for variant in enabled(32, 64, 0, 96, 12, 14, 14 | 32, 14 | 64, 6, 6 | 32, 6 | 64):
        out = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV),
                                      causal=True, _variant=variant)
        torch.cuda.synchronize()
        _check(out, ref64, ref32, torch.float16, "order variant %d" % variant)
        outs.append((variant, out.cpu()))
    for v, o in outs[1:]:      # the mapping must not change a single bit: same kernel, same per-row arithmetic (12 / 14 are other kernels)
        base = next(x for vv, x in outs if (vv & 14) == (v & 14))
        assert torch.equal(o, base), "variant %d" % v


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("Hq,Hkv", [(71, 1), (8, 2), (4, 4)], ids=["falcon7b_mqa", "gqa4", "mha"])
def test_head_dim_64(Hq, Hkv, dtype):
    """d = 64 (Falcon-7B: 71 query heads on ONE kv head): decode with fused append (auto / 1 / 3 splits) and chunked
    causal prefill over several query blocks, default and 4-wave tilings, both fragment-read paths."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(Hq)
    D, ctx, slots = 64, 1500, 4
    kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    # ---- decode ----
    B = 3
    q = torch.randn(B, 1, Hq, D).to(dtype)
    kn = torch.randn(B, 1, Hkv, D).to(dtype)
    vn = torch.randn(B, 1, Hkv, D).to(dtype)
    idx = torch.tensor([2, 0, 3], dtype=torch.int32)
    cl = torch.tensor([1, 1037, 64], dtype=torch.int32)
    kc1, vc1 = kc.clone(), vc.clone()
    ref64 = flash_attn_with_kvcache_ref(q, kc1, vc1, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
    kc2, vc2 = kc.clone(), vc.clone()
    ref32 = flash_attn_with_kvcache_ref(q, kc2, vc2, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, math="f32")
    for splits in (0, 1, 3):
        for variant in enabled(0, 1):
            kgi, vgi = kc.to(DEV), vc.to(DEV)
            out = flash_attn_with_kvcache(q.to(DEV), kgi, vgi, kn.to(DEV), vn.to(DEV), cache_seqlens=cl.to(DEV),
                                          cache_batch_idx=idx.to(DEV), causal=True, num_splits=splits, _variant=variant)
            torch.cuda.synchronize()
            _check(out, ref64, ref32, dtype, "d64 decode splits=%d variant=%d" % (splits, variant))
            assert torch.equal(kgi.cpu(), kc1) and torch.equal(vgi.cpu(), vc1)
    # ---- chunked causal prefill: 300 new rows on top of 0 / 555 cached, ragged batch ----
    n = 300
    qp = torch.randn(2, n, Hq, D).to(dtype)
    clp = torch.tensor([n, n + 555], dtype=torch.int32)
    idxp = torch.tensor([1, 3], dtype=torch.int32)
    r64 = flash_attn_with_kvcache_ref(qp, kc, vc, cache_seqlens=clp, cache_batch_idx=idxp, causal=True)
    r32 = flash_attn_with_kvcache_ref(qp, kc, vc, cache_seqlens=clp, cache_batch_idx=idxp, causal=True, math="f32")
    for variant in enabled(0, 1, 2, 8, 9, 32, 64):
        out = flash_attn_with_kvcache(qp.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=clp.to(DEV), cache_batch_idx=idxp.to(DEV),
                                      causal=True, _variant=variant)
        torch.cuda.synchronize()
        _check(out, r64, r32, dtype, "d64 prefill variant=%d" % variant)


@pytest.mark.parametrize("variant", [0, 2, 8, 1, 14, 6, 782], ids=["default", "w8", "w4", "plain_reads", "w4q2_dma_pipelined", "w8q1_dma_pipelined", "w4q2_dma_xor_image"])
@pytest.mark.parametrize("causal", [True, False], ids=["causal", "full"])
def test_prefill_kv_split(causal, variant):
    """KV-split prefill (the key range of a work item divided over workgroups, fp32 partials merged by combine_kernel):
    forced split counts incl. more splits than tiles (empty shares), ragged batch through cache_batch_idx, LSE output."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(99)
    B, n, Hq, Hkv, D, ctx = 2, 300, 8, 2, 128, 1500
    q = torch.randn(B, n, Hq, D).half()
    kc = torch.randn(3, ctx, Hkv, D).half()
    vc = torch.randn(3, ctx, Hkv, D).half()
    cl = torch.tensor([n + 900, n + 70], dtype=torch.int32)
    idx = torch.tensor([2, 0], dtype=torch.int32)
    ref64, lse64 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=causal, return_lse=True)
    ref32 = flash_attn_with_kvcache_ref(q, kc, vc, cache_seqlens=cl, cache_batch_idx=idx, causal=causal, math="f32")
    base = None
    for splits in (1, 2, 3, 5, 16):
        got = {}
        for two_launch in enabled(16384, 32768, 0):      # merged inside the launch (LAB: fence protocol, device-scope protocol), and by combine_rows_kernel (the product)
            for rep in range(2):           # twice on one stream: the merge counters reset themselves
                out, lse = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl.to(DEV), cache_batch_idx=idx.to(DEV),
                                                   causal=causal, num_splits=splits, return_softmax_lse=True, _variant=variant | two_launch)
                torch.cuda.synchronize()
                _check(out, ref64, ref32, torch.float16, "kv-split prefill splits=%d two_launch=%d" % (splits, two_launch))
                assert torch.allclose(lse.double().cpu(), lse64.double(), atol=2e-3, rtol=1e-3), "lse splits=%d" % splits
            got[two_launch] = out.float().cpu()
        assert all(torch.equal(got[0], got[m]) for m in got), "the merge forms run the same arithmetic on the same partials"
        if base is None:
            base = out.float().cpu()
        else:      # splitting only regroups the fp32 accumulation
            assert (out.float().cpu() - base).abs().max().item() <= 2e-3


def test_prefill_kv_split_heuristic_engages():
    """A tensor-parallel-shard shape (8 query heads, one kv head, a 512-token chunk on a 6 k prefix) must take the KV-split path — with
    the host-side length (workspace query > 0) AND without it: since round 4 the cache view's row count stands in for an unknown length,
    as in FlashAttention's own heuristic (flash_api.cpp:258-323), never the chunk length (rounds 1-3: no split, 105 instead of 570 TFLOP/s
    on such shapes).  The kernels divide the keys a block really sees, so the over-estimate costs balance only; results agree."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    import vattention_amd.flash_attn as FA
    torch.manual_seed(5)
    n, c, Hq, Hkv, D = 512, 6144, 8, 1, 128
    q = torch.randn(1, n, Hq, D, device=DEV).half()
    kc = torch.randn(1, 16384, Hkv, D, device=DEV).half()
    vc = torch.randn(1, 16384, Hkv, D, device=DEV).half()
    cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
    FA._workspaces.clear()
    a = flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, causal=True)
    torch.cuda.synchronize()
    assert FA._workspaces, "no host-side length: the view's 16 384 rows stand in, the key range is split"
    FA._workspaces.clear()
    b = flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, causal=True, _max_seqlen_k=c + n)
    torch.cuda.synchronize()
    assert FA._workspaces, "host-side length known: the key range is split"
    c1 = flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, causal=True, num_splits=1)      # single pass, for reference
    torch.cuda.synchronize()
    assert (a.float() - b.float()).abs().max().item() <= 2e-3 and (a.float() - c1.float()).abs().max().item() <= 2e-3


@pytest.mark.parametrize("variant,splits", [(0, 0), (2, 0), (8, 0), (0, 3), (2, 2), (14, 0), (14, 3), (6, 0), (6, 3), (782, 0), (782, 3)], ids=["default", "w8", "w4", "default_split3", "w8_split2", "dma64", "dma64_split3", "dma32", "dma32_split3", "dma64xor", "dma64xor_split3"])
def test_batched_variable_length_prefill(variant, splits):
    """One launch for the chunks of several sequences with different lengths (flash_attn_varlen_with_kvcache): every entry
    must equal the single-sequence call bit for bit and match the oracle; entries shorter than the grid's block count, an
    entry of length 1, chunks on long and short prefixes, slots through cache_batch_idx."""
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache
    torch.manual_seed(11)
    Hq, Hkv, D, ctx = 8, 2, 128, 1400
    q_lens = [300, 1, 513, 37, 256]
    prefix = [0, 700, 64, 1000, 5]
    slots = [4, 0, 2, 5, 1]
    T = sum(q_lens)
    q = torch.randn(T, Hq, D).half()
    kc = torch.randn(6, ctx, Hkv, D).half()
    vc = torch.randn(6, ctx, Hkv, D).half()
    starts = [sum(q_lens[:i]) for i in range(len(q_lens))]
    cl = [c + n for c, n in zip(prefix, q_lens)]
    qg, kg, vg = q.to(DEV), kc.to(DEV), vc.to(DEV)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=DEV)
    out = flash_attn_varlen_with_kvcache(qg, kg, vg, i32(starts), i32(q_lens), max(q_lens), i32(cl), i32(slots), causal=True,
                                         num_splits=splits, _variant=variant, _max_seqlen_k=max(cl))
    torch.cuda.synchronize()
    for i, (s0, n) in enumerate(zip(starts, q_lens)):
        qi = q[s0:s0 + n].unsqueeze(0)
        cli = torch.tensor([cl[i]], dtype=torch.int32)
        idxi = torch.tensor([slots[i]], dtype=torch.int32)
        ref64 = flash_attn_with_kvcache_ref(qi, kc, vc, cache_seqlens=cli, cache_batch_idx=idxi, causal=True)
        ref32 = flash_attn_with_kvcache_ref(qi, kc, vc, cache_seqlens=cli, cache_batch_idx=idxi, causal=True, math="f32")
        _check(out[s0:s0 + n].unsqueeze(0), ref64, ref32, torch.float16, "varlen entry %d" % i)
        if n > 1 and splits == 0:
            one = flash_attn_with_kvcache(qg[s0:s0 + n].unsqueeze(0), kg, vg, cache_seqlens=cli.to(DEV), cache_batch_idx=idxi.to(DEV),
                                          causal=True, num_splits=1, _variant=variant)
            if variant != 0:            # same tiling on both sides (the default plan picks by grid size): bit-identical
                assert torch.equal(one[0].cpu(), out[s0:s0 + n].cpu())


def test_prefill_non_causal_and_seqlen_q_gt_k():
    from vattention_amd.flash_attn import flash_attn_func, flash_attn_with_kvcache
    torch.manual_seed(5)
    q = torch.randn(2, 150, 4, 128).half()
    k = torch.randn(2, 90, 2, 128).half()
    v = torch.randn(2, 90, 2, 128).half()
    for causal in (False, True):                       # causal with Sq > Sk: leading rows fully masked -> 0
        ref64 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=90, causal=causal)
        ref32 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=90, causal=causal, math="f32")
        out = flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal)
        torch.cuda.synchronize()
        _check(out, ref64, ref32, torch.float16, "flash_attn_func causal=%s" % causal)
        if causal:
            assert torch.all(out[:, :60] == 0)


def test_online_softmax_rescale_is_exercised():
    """cdna guide rule 26: force the running max to jump at a late tile (spiked key) and check against fp64."""
    from vattention_amd.flash_attn import flash_attn_func
    torch.manual_seed(9)
    q = torch.randn(1, 256, 2, 128).half()
    k = (torch.randn(1, 700, 2, 128) * 0.3).half()
    v = torch.randn(1, 700, 2, 128).half()
    k[0, 650] = (q[0, 255, 0] * 0.5).half()            # huge score for the last query at key 650 (tile 10)
    k[0, 100] = (q[0, 200, 1] * 0.5).half()
    ref64 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=700, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=700, causal=True, math="f32")
    out = flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=True)
    torch.cuda.synchronize()
    _check(out, ref64, ref32, torch.float16, "spiked keys")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_deferred_rescale_of_the_pipelined_kernel(dtype):
    """prefill64_kernel keeps the running maximum until a tile exceeds it by more than 2^6 and then rescales O, l and the pending
    S' once (cdna guide T13 + rule 26): spiked keys at late tiles for individual rows (branch taken by one row of a wave while the
    others are not), a spike in the FIRST tile (large initial maximum: later tiles sit far below it), growth just below and just
    above the threshold, all against fp64; and the result must not depend on the threshold being crossed (same data scaled)."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(17)
    n, Lk, H = 320, 1100, 2
    q = torch.randn(1, n, H, 128).to(dtype)
    k = (torch.randn(1, Lk, H, 128) * 0.3).to(dtype)
    v = torch.randn(1, Lk, H, 128).to(dtype)
    unit = lambda x: (x.float() / x.float().norm()).to(dtype)
    k[0, 900, 0] = (unit(q[0, 319, 0]) * 6.0).to(dtype)       # one row's score jumps far above its running max at tile 14
    k[0, 5, 1] = (unit(q[0, 100, 1]) * 8.0).to(dtype)         # first tile carries a large maximum for row 100 of head 1
    k[0, 700, 1] = (unit(q[0, 250, 1]) * 3.0).to(dtype)       # moderate growth (around the 2^6 threshold after scaling)
    k[0, 701, 1] = (unit(q[0, 251, 1]) * 3.6).to(dtype)
    cl = torch.tensor([Lk], dtype=torch.int32)
    ref64 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=cl, causal=True)
    ref32 = flash_attn_with_kvcache_ref(q, k.clone(), v.clone(), cache_seqlens=cl, causal=True, math="f32")
    outs = {}
    for variant in enabled(14, 782, 0, 6):
        out = flash_attn_with_kvcache(q.to(DEV), k.to(DEV), v.to(DEV), cache_seqlens=cl.to(DEV), causal=True, _variant=variant)
        torch.cuda.synchronize()
        _check(out, ref64, ref32, dtype, "deferred rescale variant %d" % variant)
        outs[variant] = out.float().cpu()
    assert (outs[14] - outs[0]).abs().max().item() <= (4e-3 if dtype == torch.float16 else 3e-2)


def test_cache_flat_edge_cases():
    from vattention_amd.cache_ops import cache_flat
    k = torch.randn(0, 2, 128).half().to(DEV)
    kc = torch.zeros(4, 2, 128).half().to(DEV)
    cache_flat(k, k, kc, kc.clone(), "auto")            # empty input: no-op
    with pytest.raises(RuntimeError, match="Unsupported data type"):
        cache_flat(k, k, kc, kc, "fp8")
    # non-16-byte rows take the scalar kernel: head_size 20 -> 40-byte rows... kvh*hs*2 = 120 bytes
    k2 = torch.randn(9, 3, 20).half()
    v2 = torch.randn(9, 3, 20).half()
    kc2 = torch.zeros(12, 3, 20).half().to(DEV)
    vc2 = torch.zeros(12, 3, 20).half().to(DEV)
    cache_flat(k2.to(DEV), v2.to(DEV), kc2[2:], vc2[2:], "auto")
    torch.cuda.synchronize()
    assert torch.equal(kc2[2:11].cpu(), k2) and torch.equal(vc2[2:11].cpu(), v2) and torch.all(kc2[:2] == 0) and torch.all(kc2[11:] == 0)


def test_attention_on_virtual_tensors_never_touches_unmapped_pages():
    """End to end on vAttention tensors: only the prefix of each slot is mapped; the decode call passes the
    [:, :max_cache_len] view over ALL slots (wrapper :196-197).  A read past a mapped prefix would fault."""
    from vattention_amd import vattention
    from vattention_amd.cache_ops import cache_flat
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.zeros(1, device=DEV)
    mn, _ = vattention.granularity(0)
    page = 64 << 10 if (64 << 10) % mn == 0 else 2 << 20
    L, kvh, Hq, D, B, ctx = 1, 2, 8, 128, 8, 8192
    ts = vattention.init_kvcache(L, kvh, D, B, ctx, 0, torch.float16, page, False)
    try:
        K, V = ts[0], ts[1]
        vattention.reserve_physical_pages(512 * page)
        torch.manual_seed(3)
        lens = [0] * B
        seqs = {}
        for n in (700, 33, 2500):
            s = vattention.alloc_new_batch_idx(n)
            lens[s] = n
            seqs[s] = n
        vattention.step_async(lens)
        host = {}
        for s, n in seqs.items():
            k = torch.randn(n, kvh, D).half()
            v = torch.randn(n, kvh, D).half()
            host[s] = (k, v)
            kc = K[s].reshape(1, -1, kvh, D)
            vc = V[s].reshape(1, -1, kvh, D)
            cache_flat(k.to(DEV), v.to(DEV), kc.squeeze(0)[0:], vc.squeeze(0)[0:], "auto")
            q = torch.randn(1, n, Hq, D).half()
            out = flash_attn_with_kvcache(q.to(DEV), kc, vc, cache_seqlens=torch.tensor([n], dtype=torch.int32, device=DEV), causal=True)
            ref = flash_attn_with_kvcache_ref(q, k.unsqueeze(0).clone(), v.unsqueeze(0).clone(), cache_seqlens=n, causal=True)
            assert (out.double().cpu() - ref).abs().max() < 4e-3
        # one decode step for all three sequences
        slots = sorted(seqs)
        for s in slots:
            lens[s] += 1
        vattention.step_async(lens)
        q = torch.randn(len(slots), 1, Hq, D).half()
        kn = torch.randn(len(slots), 1, kvh, D).half()
        vn = torch.randn(len(slots), 1, kvh, D).half()
        cl = torch.tensor([seqs[s] for s in slots], dtype=torch.int32)
        mx = int(cl.max()) + 1
        out = flash_attn_with_kvcache(q.to(DEV), K[:, :mx], V[:, :mx], kn.to(DEV), vn.to(DEV), cache_seqlens=cl.to(DEV),
                                      cache_batch_idx=torch.tensor(slots, dtype=torch.int32, device=DEV), causal=True)
        torch.cuda.synchronize()
        for i, s in enumerate(slots):
            kf = torch.cat([host[s][0], kn[i]], 0).unsqueeze(0)
            vf = torch.cat([host[s][1], vn[i]], 0).unsqueeze(0)
            ref = flash_attn_with_kvcache_ref(q[i:i + 1], kf, vf, cache_seqlens=seqs[s] + 1)
            assert (out[i:i + 1].double().cpu() - ref).abs().max() < 4e-3
    finally:
        vattention.cleanup()


def test_decode_call_is_hip_graph_capturable():
    """An engine that captures its decode step in a HIP graph (torch.cuda.graph) can include this path: the launch makes no
    allocation or synchronising call once the split-KV workspace exists, and everything that changes between steps (lengths,
    slots, the new K/V rows, q) is read from device memory at replay time."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    torch.manual_seed(21)
    B, Hq, Hkv, D, ctx = 4, 8, 2, 128, 3000
    kc = torch.randn(6, ctx, Hkv, D, device=DEV).half()
    vc = torch.randn(6, ctx, Hkv, D, device=DEV).half()
    q = torch.randn(B, 1, Hq, D, device=DEV).half()
    kn = torch.randn(B, 1, Hkv, D, device=DEV).half()
    vn = torch.randn(B, 1, Hkv, D, device=DEV).half()
    cl = torch.tensor([100, 2500, 31, 1999], dtype=torch.int32, device=DEV)
    idx = torch.tensor([5, 0, 3, 1], dtype=torch.int32, device=DEV)
    out = torch.empty_like(q)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up on the capture stream: creates that stream's workspace
        flash_attn_with_kvcache(q, kc.clone(), vc.clone(), kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    kg, vg = kc.clone(), vc.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        flash_attn_with_kvcache(q, kg, vg, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True, out=out)
    for step in range(3):
        # new inputs written into the captured buffers, then replay
        q.copy_(torch.randn_like(q)); kn.copy_(torch.randn_like(kn)); vn.copy_(torch.randn_like(vn))
        if step:
            cl.add_(1)
        ke, ve = kg.clone(), vg.clone()
        g.replay()
        torch.cuda.synchronize()
        ref = flash_attn_with_kvcache(q, ke, ve, kn, vn, cache_seqlens=cl, cache_batch_idx=idx, causal=True)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), step
        assert torch.equal(kg, ke) and torch.equal(vg, ve)


@pytest.mark.lab
@pytest.mark.parametrize("sel,sub", [(1, 0), (5, 0), (6, 0), (7, 13), (7, 4)], ids=["vt_preread", "scalar_scale", "mov_fmac", "soffset_dma", "price_list_v_mov"])
def test_prefill64_round6_lab_builds_are_bit_identical_to_the_product(sel, sub):
    """The exact-arithmetic builds of round 6's issue-budget study (tools/lab/csrc/prefill64_lab.hip, variant bits 28-30 + the sub-selector in
    split_reserved bits 16-23) run the product's arithmetic in the product's order: bit-identical outputs, on a causal chunk over a prefix and on a
    ragged whole prompt.  (The product kernel carries the winners: V^T pre-read, scalar scale, DMA distances in the scalar offset.)"""
    import ctypes as C
    from tools.kbench import params
    from vattention_amd import kernels as K
    st = torch.cuda.current_stream().cuda_stream
    for n, c, Hq, Hkv in ((3000, 500, 8, 2), (777, 1000, 4, 4)):
        torch.manual_seed(n)
        q = torch.randn(1, n, Hq, 128, device=DEV, dtype=torch.float16)
        kc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        vc = torch.randn(1, c + n, Hkv, 128, device=DEV, dtype=torch.float16)
        cl = torch.tensor([c + n], dtype=torch.int32, device=DEV)
        outs = []
        for v, s_ in ((14, 0), (14 | (sel << 28), sub)):
            p, keep = params(q, kc, vc, cl, variant=v)
            p.split_reserved |= s_ << 16
            lib = K.klib_for(v)
            assert lib.vattn_flash_attn_with_kvcache(C.byref(p), C.c_void_p(st)) == 0, K.last_error(lib)
            torch.cuda.synchronize()
            outs.append(keep[0].clone())
        assert torch.equal(outs[0], outs[1]), "lab build %d.%d differs from the product: max |diff| %.3e" % (sel, sub, (outs[0].float() - outs[1].float()).abs().max().item())
