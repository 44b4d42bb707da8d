"""prefill64p_kernel (round 5): work lists walked by PERSISTENT workgroups — one per CU, each a queue of (entry, head, query block,
key-tile range) pieces whose K / V tile stream does not stop between pieces — against the oracle, against the one-workgroup-per-piece
launch of the same pieces (bit for bit), with few long queues (every seam kind: pieces of one, two, three and many tiles, pieces
without tiles, partial and direct outputs following each other), stale host lengths, strided cache views and the varlen form; both
queue forms: host-assigned (`drawn=False`: the product default, flash_attn.PERSISTENT_DRAWN = False) and drawn from the device counter
(`drawn=True`), whose counters must be back at zero after every launch."""
import ctypes as C

import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(out, ref64, ref32, dtype, what):
    o = out.float().cpu()
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    err = (o - ref64.float()).abs().max().item()
    err32 = (ref32.float() - ref64.float()).abs().max().item()
    assert torch.allclose(o, ref64.float(), atol=tol, rtol=tol), "%s: max err %.3g" % (what, err)
    assert err <= 2 * err32 + 1e-5 + (0 if dtype == torch.float16 else 8e-3), "%s: err %.3g vs reference-numerics err %.3g" % (what, err, err32)


def _run(chunks, Hq, Hkv, dtype, plans, causal=True, seed=5, klens_dev=None, strided=False, lse=False):
    """chunks: [(cached, new)] per entry.  plans: [(name, kwargs for FA.prefill_plan | None for the default launch)].  Returns
    {name: output}, reference pair."""
    from vattention_amd import flash_attn as FA
    from vattention_amd import kernels as K
    torch.manual_seed(seed)
    D, P = 128, len(chunks)
    ctx = max(c + n for c, n in chunks) + 70
    slots = P + 2
    if strided:      # one layer's view of a megacache-like tensor: rows 3 x Hkv x D elements apart
        kc = torch.randn(slots, ctx, 3, Hkv, D).to(dtype)[:, :, 1]
        vc = torch.randn(slots, ctx, 3, Hkv, D).to(dtype)[:, :, 2]
    else:
        kc = torch.randn(slots, ctx, Hkv, D).to(dtype)
        vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    T = sum(n for _, n in chunks)
    q = torch.randn(T, Hq, D).to(dtype)
    sl = torch.randperm(slots)[:P].to(torch.int32)
    q_lens, k_lens = [n for _, n in chunks], [c + n for c, n in chunks]
    dev_lens = klens_dev if klens_dev is not None else k_lens
    refs64, refs32, tok = [], [], 0
    for i, (c, n) in enumerate(chunks):
        s_ = int(sl[i])
        for math, dst in (({}, refs64), ({"math": "f32"}, refs32)):
            dst.append(flash_attn_with_kvcache_ref(q[tok:tok + n].unsqueeze(0), kc[s_:s_ + 1].clone(), vc[s_:s_ + 1].clone(),
                                                   cache_seqlens=torch.tensor([dev_lens[i]], dtype=torch.int32), causal=causal, **math)[0])
        tok += n
    ref64, ref32 = torch.cat(refs64), torch.cat(refs32)
    kg, vg, qg = kc.to(DEV), vc.to(DEV), q.to(DEV)
    if strided:
        kfull = torch.randn(slots, ctx, 3, Hkv, D).to(dtype).to(DEV)
        vfull = torch.randn(slots, ctx, 3, Hkv, D).to(dtype).to(DEV)
        kfull[:, :, 1] = kg
        vfull[:, :, 2] = vg
        kg, vg = kfull[:, :, 1], vfull[:, :, 2]
    p = K.AttnParams()
    p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = P, max(q_lens), Hq, Hkv, D, int(causal)
    p.o_row_stride, p.o_head_stride = Hq * D, D
    outs, info = {}, {}
    for name, kw in plans:
        pl = FA.prefill_plan(p, q_lens, k_lens, torch.device(DEV), **kw) if kw is not None else None
        if kw is not None:
            assert pl.t is not None, "%s: no work list" % name
            assert (pl.n_wg > 0) == bool(kw.get("persistent", True)), name
            info[name] = (pl.n_items, pl.n_blocks, pl.n_wg)
        out = torch.full((T, Hq, D), float("nan"), dtype=dtype, device=DEV)
        starts = torch.tensor([sum(q_lens[:i]) for i in range(P)], dtype=torch.int32, device=DEV)
        FA.flash_attn_varlen_with_kvcache(qg, kg, vg, starts, torch.tensor(q_lens, dtype=torch.int32, device=DEV), max(q_lens),
                                          torch.tensor(dev_lens, dtype=torch.int32, device=DEV), sl.to(DEV), causal=causal, out=out,
                                          _max_seqlen_k=max(k_lens), _pf_plan=pl)
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any(), "%s: rows left unwritten" % name
        outs[name] = out
    return outs, ref64, ref32, info


SHAPES = [
    ("tp8_8k", 8, 1, [(0, 8192)]),                                       # one underfilled round: long blocks cut, partials merged
    ("llama8b_3prompts", 32, 8, [(0, 2300), (0, 4119), (0, 700)]),       # ragged batch, several pieces per queue
    ("chunk_on_prefix", 8, 2, [(6000, 1000)]),
    ("gqa7_ragged_rows", 14, 2, [(0, 1500), (100, 333)]),                # G = 7: one XCD class; ragged last blocks
    ("mha_short", 4, 4, [(0, 130), (0, 64), (0, 2)]),                    # pieces of 1-3 tiles: every step is padded or a seam
]


# (the two largest shapes' oracle runs take 15-25 s each: bf16 rides on the three smaller ones, whose queues hold every seam kind)
@pytest.mark.parametrize("name,Hq,Hkv,chunks,dtype", [(s[0], s[1], s[2], s[3], dt) for s in SHAPES for dt in (torch.float16, torch.bfloat16)
                                                      if not (dt == torch.bfloat16 and s[0] in ("tp8_8k", "llama8b_3prompts"))],
                         ids=["%s-%s" % (s[0], "f16" if dt == torch.float16 else "bf16") for s in SHAPES for dt in (torch.float16, torch.bfloat16)
                              if not (dt == torch.bfloat16 and s[0] in ("tp8_8k", "llama8b_3prompts"))])
def test_persistent_work_list_matches_the_oracle_and_the_per_piece_launch(name, Hq, Hkv, chunks, dtype):
    plans = [("persistent", dict(persistent=True)), ("drawn_t9", dict(persistent=True, force_tiles=9, drawn=True)),
             ("assigned_t9", dict(persistent=True, force_tiles=9, drawn=False)), ("drawn_t9_again", dict(persistent=True, force_tiles=9, drawn=True)),
             ("per_piece_t9", dict(persistent=False, force_tiles=9)), ("default", None)]
    outs, ref64, ref32, info = _run(chunks, Hq, Hkv, dtype, plans)
    for k, o in outs.items():
        _check(o, ref64, ref32, dtype, "%s / %s %s" % (name, k, info.get(k)))
    # the SAME pieces through both kernels, whoever walks them in whatever order: same tile arithmetic in the same order inside a piece
    # (a padded step adds P = 0), same merge.  The second drawn launch finds the counters the first one left: zero.
    assert info["drawn_t9"][:2] == info["per_piece_t9"][:2] == info["assigned_t9"][:2]
    for k in ("drawn_t9", "assigned_t9", "drawn_t9_again"):
        assert torch.equal(outs[k], outs["per_piece_t9"]), "%s: %s and per-piece launches of one list differ" % (name, k)


@pytest.mark.parametrize("max_wg", [1, 8, 24])
@pytest.mark.parametrize("tiles", [1, 2, 3, 5])
def test_long_queues_of_short_pieces(max_wg, tiles):
    """Few workgroups, many pieces each, cut to `tiles` key tiles: one-, two- and three-tile pieces are stepped over three tiles with the
    tiles behind their own masked whole; every piece but a queue's first starts at a seam."""
    chunks = [(0, 700), (300, 260), (0, 1), (0, 1030)]
    plans = [("q", dict(persistent=True, force_tiles=tiles, max_wg=max_wg, drawn=True)), ("qa", dict(persistent=True, force_tiles=tiles, max_wg=max_wg, drawn=False)),
             ("q2", dict(persistent=True, force_tiles=tiles, max_wg=max_wg, drawn=True)), ("ref_list", dict(persistent=False, force_tiles=tiles))]
    outs, ref64, ref32, info = _run(chunks, 4, 2, torch.float16, plans, seed=tiles * 31 + max_wg)
    assert info["q"][2] == min(max_wg, info["q"][0]) or info["q"][2] % 8 == 0
    _check(outs["q"], ref64, ref32, torch.float16, "queues of %d, pieces of %d tiles %s" % (max_wg, tiles, info["q"]))
    for k in ("q", "qa", "q2"):
        assert torch.equal(outs[k], outs["ref_list"]), k


def test_non_causal_and_strided_views():
    for causal, strided in ((False, False), (True, True), (False, True)):
        plans = [("p", dict(persistent=True, force_tiles=7, max_wg=16)), ("pa", dict(persistent=True, force_tiles=7, max_wg=16, drawn=False)),
                 ("l", dict(persistent=False, force_tiles=7))]
        outs, ref64, ref32, info = _run([(500, 300), (0, 900)], 8, 2, torch.float16, plans, causal=causal, strided=strided, seed=11)
        _check(outs["p"], ref64, ref32, torch.float16, "causal=%s strided=%s %s" % (causal, strided, info["p"]))
        assert torch.equal(outs["p"], outs["l"]) and torch.equal(outs["pa"], outs["l"])


@pytest.mark.parametrize("delta", [-900, -64, -1, 70], ids=["much_shorter", "one_tile_shorter", "one_key_shorter", "longer"])
def test_stale_host_lengths_cost_balance_never_keys(delta):
    """The list is a hint (include/vattn_kernels.h): built from host lengths that differ from cache_seqlens on the device, every piece
    is clamped to what its block really sees — pieces that lose all their tiles included (stepped over masked tiles: zeros / -inf)."""
    chunks = [(1000, 520), (0, 1300)]
    k_host = [c + n for c, n in chunks]
    k_dev = [max(n, kl + delta) for kl, (c, n) in zip(k_host, chunks)]
    k_dev = [min(kd, kl + 70) for kd, kl in zip(k_dev, k_host)]
    plans = [("p", dict(persistent=True, force_tiles=4, max_wg=8)), ("pa", dict(persistent=True, force_tiles=4, max_wg=8, drawn=False)), ("d", None)]
    outs, ref64, ref32, info = _run(chunks, 8, 2, torch.float16, plans, klens_dev=k_dev, seed=3)
    for k in ("p", "pa"):
        _check(outs[k], ref64, ref32, torch.float16, "%s: host lengths %s, device lengths %s %s" % (k, k_host, k_dev, info[k]))


def test_plan_describe_reports_the_persistent_launch():
    from vattention_amd import flash_attn as FA
    from vattention_amd import kernels as K
    p = K.AttnParams()
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d, p.is_causal = 1, 8192, 8192, 8, 1, 128, 1
    p.o_row_stride, p.o_head_stride = 8 * 128, 128
    pl = FA.prefill_plan(p, None, [8192], torch.device(DEV), persistent=True)
    assert pl.t is not None and pl.n_wg == 256
    pl.attach(p)
    d = K.describe(p)
    assert d["path"] == 1 and d["tiling"] == 7 and d["workgroups"] == 256 and d["merge_launch"] == int(pl.n_blocks > 0)


def test_default_policy_takes_persistent_workgroups_for_a_chunk_on_a_long_prefix_only():
    """flash_attn.PERSISTENT = "chunks" (measured: profiles/r05_p64p_*): one entry whose prefix is at least four times its chunk gets the
    persistent queues; whole prompts and ragged batches keep one workgroup per piece."""
    from vattention_amd import flash_attn as FA
    from vattention_amd import kernels as K
    assert FA.PERSISTENT == "chunks"

    def plan(q, k):
        p = K.AttnParams()
        p.b, p.seqlen_q, p.h, p.h_k, p.d, p.is_causal = len(q), max(q), 8, 1, 128, 1
        return FA.prefill_plan(p, q, k, torch.device(DEV))
    assert plan([2048], [32768]).n_wg > 0                    # Sarathi chunk on a prefix
    assert plan([8192], [8192]).n_wg == 0                    # whole prompt
    assert plan([2048, 2048], [32768, 32768]).n_wg == 0      # batch
