"""The reference docstring's literal examples (flash_attn_interface.py:1187-1203) asserted against the HIP kernels through the C ABI
(vattention_amd.flash_attn -> libvattn_amd.so): the 2 x 5 and 5 x 2 bottom-right causal masks, all-zero mask rows -> 0, and
"6 query heads over 2 kv heads: heads 0-2 -> kv head 0, heads 3-5 -> kv head 1".  Probes and expected matrices are those of the CPU
test (tests/test_docstring_pins.py); head dimensions 64 and 128 reach prefill_kernel and prefill64_kernel, seqlen_q = 1 the decode form."""
import pytest
import torch

from tests.test_docstring_pins import MASK_2x5, MASK_5x2, gqa_probe, mask_probe, read_mask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(q, k, v, seqlen_k):
    from vattention_amd.flash_attn import flash_attn_with_kvcache
    # the cache view is longer than the visible keys (rows beyond cache_seqlens hold garbage the operator must not read)
    kc = torch.full((1, seqlen_k + 3, k.shape[2], k.shape[3]), float("nan"), dtype=k.dtype)
    vc = torch.full_like(kc, float("nan"))
    kc[:, :seqlen_k], vc[:, :seqlen_k] = k, v
    cl = torch.tensor([seqlen_k], dtype=torch.int32, device=DEV)
    out = flash_attn_with_kvcache(q.to(DEV), kc.to(DEV), vc.to(DEV), cache_seqlens=cl, causal=True)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("D", [64, 128])
def test_docstring_mask_2x5_on_the_kernels(D, dtype):
    q, k, v = mask_probe(2, 5, D, dtype)
    o = _run(q, k, v, 5)
    assert read_mask(o, 5) == MASK_2x5
    assert torch.allclose(o[0, 0, 0, :5].double(), torch.tensor([0.25, 0.25, 0.25, 0.25, 0.0], dtype=torch.float64), atol=2e-3)
    assert torch.allclose(o[0, 1, 0, :5].double(), torch.full((5,), 0.2, dtype=torch.float64), atol=2e-3)
    assert bool((o[0, :, 0, 5:] == 0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("D", [64, 128])
def test_docstring_mask_5x2_and_all_zero_rows_on_the_kernels(D, dtype):
    q, k, v = mask_probe(5, 2, D, dtype)
    o = _run(q, k, v, 2)
    assert read_mask(o, 2) == MASK_5x2
    assert bool((o[0, :3] == 0).all())            # flash_attn_interface.py:1203: an all-zero mask row gives a zero output row
    assert float(o[0, 3, 0, 0]) == 1.0 and torch.allclose(o[0, 4, 0, :2].double(), torch.full((2,), 0.5, dtype=torch.float64), atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("Sq,Sk,D", [(1, 9, 128), (1, 700, 128), (4, 9, 128), (300, 300, 128), (4, 9, 64), (1, 9, 64)])
def test_docstring_gqa_6_heads_over_2_on_the_kernels(Sq, Sk, D, dtype):
    q, k, v = gqa_probe(Sq=Sq, Sk=Sk, D=D, dtype=dtype)
    o = _run(q, k, v, Sk).double()
    want = torch.tensor([1.0, 1.0, 1.0, 2.0, 2.0, 2.0], dtype=torch.float64).view(1, 1, 6, 1).expand_as(o)
    # a convex combination of identical rows: exact up to the rounding of the normalisation
    assert torch.allclose(o, want, atol=4e-3 if dtype == torch.float16 else 2e-2, rtol=0)
