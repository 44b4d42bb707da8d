"""Fused prefill || decode launch (vattn_hybrid_attn) against the CPU oracle — not against the serial path: the
prefill part (one chunk on a cache prefix, or several chunks of different lengths) and the decode part (batch with in-kernel
append, cache_batch_idx, ragged contexts) of a hybrid iteration, each compared with oracle/attn.py on the same inputs; appended
K/V rows bit-exact; back-to-back launches (the control words reset themselves); all three role policies."""
import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref
from tests.test_gpu_attention import _check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(dtype, Hq, Hkv, chunks, dec_lens, seed):
    """chunks: [(cache_len, q_len)] prefill entries; dec_lens: context of every decode sequence.  Slots are disjoint."""
    torch.manual_seed(seed)
    D = 128
    P, Bd = len(chunks), len(dec_lens)
    ctx = max([c + n for c, n in chunks] + [l + 1 for l in dec_lens]) + 5
    slots = P + Bd + 2
    kc = (torch.randn(slots, ctx, Hkv, D) * 0.7).to(dtype)
    vc = torch.randn(slots, ctx, Hkv, D).to(dtype)
    perm = torch.randperm(slots)
    p_slots, d_slots = perm[:P].to(torch.int32), perm[P:P + Bd].to(torch.int32)
    T = sum(n for _, n in chunks)
    q = torch.randn(T + Bd, Hq, D).to(dtype)
    kn = torch.randn(Bd, 1, Hkv, D).to(dtype)
    vn = torch.randn(Bd, 1, Hkv, D).to(dtype)
    return dict(D=D, P=P, Bd=Bd, ctx=ctx, kc=kc, vc=vc, p_slots=p_slots, d_slots=d_slots, T=T, q=q, kn=kn, vn=vn)


def _oracle(c, chunks, dec_lens, dtype, math):
    outs = []
    tok = 0
    for i, (cl, n) in enumerate(chunks):
        s = int(c["p_slots"][i])
        o = flash_attn_with_kvcache_ref(c["q"][tok:tok + n].unsqueeze(0), c["kc"][s:s + 1].clone(), c["vc"][s:s + 1].clone(),
                                        cache_seqlens=torch.tensor([cl + n], dtype=torch.int32), causal=True, **math)
        outs.append(o[0])
        tok += n
    kcd, vcd = c["kc"].clone(), c["vc"].clone()
    ml = max(dec_lens) + 1
    od = flash_attn_with_kvcache_ref(c["q"][tok:].unsqueeze(1), kcd[:, :ml], vcd[:, :ml], c["kn"], c["vn"],
                                     cache_seqlens=torch.tensor(dec_lens, dtype=torch.int32), cache_batch_idx=c["d_slots"], causal=True, **math)
    return torch.cat(outs + [od[:, 0]], 0), kcd, vcd


@pytest.mark.lab      # the FUSED launch is a lab kernel (tools/lab/csrc/hybrid_lab.hip)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("Hq,Hkv,chunks,dec_lens", [
    (8, 2, [(1000, 512)], [700, 33, 1500, 1, 257]),                       # Sarathi iteration: one chunk + running decodes
    (32, 8, [(0, 300), (640, 129), (64, 1)], [900, 2000]),                # several chunks of different lengths (varlen form)
    (8, 1, [(2048, 1024)], [3000] * 9),                                   # TP8 shard heads, G = 8, equal decode contexts
    (14, 2, [(100, 200)], [64, 4000, 31]),                                # G = 7
], ids=["chunk512_dec5", "varlen3_dec2", "tp8_dec9", "g7"])
def test_fused_hybrid_launch_matches_the_oracle(Hq, Hkv, chunks, dec_lens, dtype):
    from vattention_amd.flash_attn import flash_attn_varlen_with_kvcache, flash_attn_with_kvcache, hybrid_attn
    c = _case(dtype, Hq, Hkv, chunks, dec_lens, 11)
    ref64, kc_ref, vc_ref = _oracle(c, chunks, dec_lens, dtype, {})
    ref32, _, _ = _oracle(c, chunks, dec_lens, dtype, {"math": "f32"})
    P, Bd, T, D = c["P"], c["Bd"], c["T"], c["D"]
    for role_mode, rounds, dvar in ((0, 3, 0), (1, 1, 0), (2, 1, 0), (0, 3, 1024)):      # dvar 1024: split merge through device-scope accesses
        kg, vg = c["kc"].to(DEV), c["vc"].to(DEV)
        q = c["q"].to(DEV)
        for _ in range(rounds):           # the same launch again: idempotent (the append rewrites the same rows), control words self-reset
            out = torch.full((T + Bd, Hq, D), float("nan"), dtype=dtype, device=DEV)
            starts = torch.tensor([sum(n for _, n in chunks[:i]) for i in range(P)], dtype=torch.int32, device=DEV)
            qlens = torch.tensor([n for _, n in chunks], dtype=torch.int32, device=DEV)
            totals = torch.tensor([cl + n for cl, n in chunks], dtype=torch.int32, device=DEV)
            if P == 1:
                s = int(c["p_slots"][0])
                pre = lambda: flash_attn_with_kvcache(q[:T].unsqueeze(0), kg[s:s + 1], vg[s:s + 1], cache_seqlens=totals, causal=True,
                                                      out=out[:T].unsqueeze(0), _max_seqlen_k=chunks[0][0] + chunks[0][1])
            else:
                pre = lambda: flash_attn_varlen_with_kvcache(q[:T], kg, vg, starts, qlens, max(n for _, n in chunks), totals,
                                                             c["p_slots"].to(DEV), causal=True, out=out[:T],
                                                             _max_seqlen_k=max(cl + n for cl, n in chunks))
            ml = max(dec_lens) + 1
            dec = lambda: flash_attn_with_kvcache(q[T:].unsqueeze(1), kg[:, :ml], vg[:, :ml], c["kn"].to(DEV), c["vn"].to(DEV),
                                                  cache_seqlens=torch.tensor(dec_lens, dtype=torch.int32, device=DEV),
                                                  cache_batch_idx=c["d_slots"].to(DEV), causal=True, out=out[T:].unsqueeze(1), _variant=dvar)
            hybrid_attn(pre, dec, torch.device(DEV), _role_mode=role_mode)
            torch.cuda.synchronize()
            assert not torch.isnan(out.float()).any(), "rows left unwritten (role mode %d)" % role_mode
            _check(out, ref64, ref32, dtype, "fused hybrid launch, role mode %d" % role_mode)
        assert torch.equal(kg.cpu(), kc_ref) and torch.equal(vg.cpu(), vc_ref), "appended rows differ"


@pytest.mark.lab      # the FUSED launch is a lab kernel (tools/lab/csrc/hybrid_lab.hip)
def test_fused_hybrid_workspace_survives_a_growing_decode_batch():
    """Back-to-back launches on ONE stream reuse one workspace.  A first launch with a few decode groups leaves fp32 split partials
    in it; a later launch with MORE groups (b * h_k going from 8 to 96 here) must not find its merge counters on top of those bytes
    — they live in a fixed-capacity region ahead of the partials (csrc/hybrid_kernels.hip, HY_DONE_CAP).  With the old
    batch-dependent offset the larger launch's counters started non-zero, the merge never ran, and decode rows stayed unwritten."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache, hybrid_attn
    dtype, Hq, Hkv = torch.float16, 8, 2
    # largest batch first (sizes the workspace once), then a small one (leaves partials right behind ITS counters), then a larger one
    for dec_lens in ([300 + 11 * i for i in range(130)], [900, 1800, 2500, 700], [600 + 37 * i for i in range(48)], [1500] * 3,
                     [400 + 5 * i for i in range(100)]):
        chunks = [(512, 256)]
        c = _case(dtype, Hq, Hkv, chunks, dec_lens, 23 + len(dec_lens))
        ref64, kc_ref, vc_ref = _oracle(c, chunks, dec_lens, dtype, {})
        ref32, _, _ = _oracle(c, chunks, dec_lens, dtype, {"math": "f32"})
        T, Bd, D = c["T"], c["Bd"], c["D"]
        kg, vg, q = c["kc"].to(DEV), c["vc"].to(DEV), c["q"].to(DEV)
        out = torch.full((T + Bd, Hq, D), float("nan"), dtype=dtype, device=DEV)
        s = int(c["p_slots"][0])
        totals = torch.tensor([chunks[0][0] + chunks[0][1]], dtype=torch.int32, device=DEV)
        ml = max(dec_lens) + 1
        hybrid_attn(lambda: flash_attn_with_kvcache(q[:T].unsqueeze(0), kg[s:s + 1], vg[s:s + 1], cache_seqlens=totals, causal=True,
                                                    out=out[:T].unsqueeze(0), _max_seqlen_k=768),
                    lambda: flash_attn_with_kvcache(q[T:].unsqueeze(1), kg[:, :ml], vg[:, :ml], c["kn"].to(DEV), c["vn"].to(DEV),
                                                    cache_seqlens=torch.tensor(dec_lens, dtype=torch.int32, device=DEV),
                                                    cache_batch_idx=c["d_slots"].to(DEV), causal=True, out=out[T:].unsqueeze(1)),
                    torch.device(DEV))
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any(), "decode rows left unwritten with %d sequences" % Bd
        _check(out, ref64, ref32, dtype, "fused hybrid launch, %d decode sequences after a smaller batch" % Bd)
        assert torch.equal(kg.cpu(), kc_ref) and torch.equal(vg.cpu(), vc_ref)


@pytest.mark.lab      # the FUSED launch is a lab kernel (tools/lab/csrc/hybrid_lab.hip)
def test_fused_hybrid_argument_rules():
    from vattention_amd.flash_attn import flash_attn_with_kvcache, hybrid_attn
    dt = torch.float16
    q = torch.randn(1, 64, 4, 128, device=DEV, dtype=dt)
    kc = torch.randn(2, 256, 2, 128, device=DEV, dtype=dt)
    qd = torch.randn(1, 1, 4, 128, device=DEV, dtype=dt)
    cl = torch.tensor([100], dtype=torch.int32, device=DEV)
    pre = lambda: flash_attn_with_kvcache(q, kc[:1], kc[:1], cache_seqlens=cl, causal=True)
    dec = lambda: flash_attn_with_kvcache(qd, kc[1:], kc[1:], cache_seqlens=cl, causal=True)
    with pytest.raises(RuntimeError, match="first part must be a prefill"):
        hybrid_attn(dec, dec, torch.device(DEV))
    with pytest.raises(RuntimeError, match="exactly one attention call"):
        hybrid_attn(lambda: None, dec, torch.device(DEV))
    q64 = torch.randn(1, 64, 4, 64, device=DEV, dtype=dt)
    k64 = torch.randn(2, 256, 2, 64, device=DEV, dtype=dt)
    with pytest.raises(RuntimeError, match="head dimension 128"):
        hybrid_attn(lambda: flash_attn_with_kvcache(q64, k64[:1], k64[:1], cache_seqlens=cl, causal=True),
                    lambda: flash_attn_with_kvcache(q64[:, :1], k64[1:], k64[1:], cache_seqlens=cl, causal=True), torch.device(DEV))
    hybrid_attn(pre, dec, torch.device(DEV))          # and the valid form runs
    torch.cuda.synchronize()


def test_product_entry_point_issues_the_two_launches_back_to_back():
    """Round 4: the fused kernel is lab-only; the PRODUCT library's vattn_hybrid_attn (what a binding of the reference's POD call
    site reaches) runs the plan-chosen prefill launch and the device-planned decode launch in order on the caller's stream — results
    bit-identical to the two stand-alone calls, appended rows in place."""
    from vattention_amd.flash_attn import flash_attn_with_kvcache, hybrid_attn
    torch.manual_seed(5)
    Hq, Hkv, D, ctx = 8, 2, 128, 3000
    kc = torch.randn(6, ctx, Hkv, D, device=DEV).half()
    vc = torch.randn(6, ctx, Hkv, D, device=DEV).half()
    T, c = 700, 1500
    q = torch.randn(T + 4, Hq, D, device=DEV).half()
    kn, vn = torch.randn(4, 1, Hkv, D, device=DEV).half(), torch.randn(4, 1, Hkv, D, device=DEV).half()
    totals = torch.tensor([c + T], dtype=torch.int32, device=DEV)
    dl = torch.tensor([2500, 31, 900, 1777], dtype=torch.int32, device=DEV)
    di = torch.tensor([1, 2, 4, 5], dtype=torch.int32, device=DEV)

    def run(fused_entry):
        kg, vg = kc.clone(), vc.clone()
        out = torch.zeros_like(q)
        pre = lambda: flash_attn_with_kvcache(q[:T].unsqueeze(0), kg[0:1], vg[0:1], cache_seqlens=totals, causal=True, out=out[:T].unsqueeze(0))
        dec = lambda: flash_attn_with_kvcache(q[T:].unsqueeze(1), kg[:, :2501], vg[:, :2501], kn, vn, cache_seqlens=dl, cache_batch_idx=di, causal=True,
                                              out=out[T:].unsqueeze(1))
        if fused_entry:
            hybrid_attn(pre, dec, torch.device(DEV), _product=True)
        else:
            pre()
            dec()
        torch.cuda.synchronize()
        return out, kg, vg

    a, ka, va_ = run(True)
    b, kb, vb = run(False)
    assert torch.equal(a, b) and torch.equal(ka, kb) and torch.equal(va_, vb)
    assert float(a.abs().max()) > 0
