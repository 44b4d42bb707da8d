"""The two statements the reference's own docstring makes about flash_attn_with_kvcache with literal examples
(/root/reference/pod_attn/pod_attn/flash_attn_interface.py:1187-1203), asserted literally against the CPU oracle:

  * GQA: "if Q has 6 heads and K, V have 2 heads, head 0, 1, 2 of Q will attention to head 0 of K, V, and head 3, 4, 5 of Q will
    attention to head 1 of K, V"  (:1187-1190);
  * bottom-right causal alignment: the 2 x 5 and the 5 x 2 keep / mask matrices (:1192-1202), and "If the row of the mask is all
    zero, the output will be zero" (:1203).

The mask is READ BACK from the operator's output: with q = 0 every visible key gets the same probability, and with one-hot value rows
(v[j] = e_j) output element j of query i is 1 / (visible keys of i) if key j is visible and exactly 0 if it is not.
The same assertions run against the HIP kernels in tests/test_gpu_docstring_pins.py."""
import pytest
import torch

from oracle.attn import flash_attn_with_kvcache_ref

MASK_2x5 = [[1, 1, 1, 1, 0],
            [1, 1, 1, 1, 1]]                # flash_attn_interface.py:1193-1195
MASK_5x2 = [[0, 0], [0, 0], [0, 0], [1, 0], [1, 1]]      # :1196-1202


def mask_probe(seqlen_q, seqlen_k, D, dtype=torch.float16, heads=1):
    """inputs whose output IS the normalised mask: q = 0 (uniform softmax over the visible keys), v[j] = e_j"""
    q = torch.zeros(1, seqlen_q, heads, D, dtype=dtype)
    k = torch.randn(1, seqlen_k, heads, D).to(dtype)
    v = torch.zeros(1, seqlen_k, heads, D, dtype=dtype)
    for j in range(seqlen_k):
        v[0, j, :, j] = 1.0
    return q, k, v


def read_mask(out, seqlen_k):
    """keep / mask matrix [seqlen_q][seqlen_k] from an output of mask_probe inputs (head 0)"""
    return [[int(x > 0) for x in row[:seqlen_k]] for row in out[0, :, 0].double().tolist()]


def gqa_probe(Hq=6, Hkv=2, Sq=1, Sk=9, D=16, dtype=torch.float16):
    """every value row of kv head g is the constant g + 1: whatever the scores, a query head's output is (its kv head + 1)"""
    torch.manual_seed(5)
    q = torch.randn(1, Sq, Hq, D).to(dtype)
    k = torch.randn(1, Sk, Hkv, D).to(dtype)
    v = torch.empty(1, Sk, Hkv, D, dtype=dtype)
    for g in range(Hkv):
        v[0, :, g] = float(g + 1)
    return q, k, v


@pytest.mark.parametrize("math", ["f64", "f32"])
def test_docstring_mask_2x5(math):
    q, k, v = mask_probe(2, 5, 16)
    o = flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=5, causal=True, math=math)
    assert read_mask(o, 5) == MASK_2x5
    assert torch.allclose(o[0, 0, 0, :5].double(), torch.tensor([0.25, 0.25, 0.25, 0.25, 0.0], dtype=torch.float64), atol=1e-3)
    assert torch.allclose(o[0, 1, 0, :5].double(), torch.full((5,), 0.2, dtype=torch.float64), atol=1e-3)


@pytest.mark.parametrize("math", ["f64", "f32"])
def test_docstring_mask_5x2_and_all_zero_rows(math):
    q, k, v = mask_probe(5, 2, 16)
    o = flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=2, causal=True, math=math)
    assert read_mask(o, 2) == MASK_5x2
    assert bool((o[0, :3] == 0).all())            # ":1203 If the row of the mask is all zero, the output will be zero"


@pytest.mark.parametrize("Sq", [1, 4])
def test_docstring_gqa_6_heads_over_2(Sq):
    q, k, v = gqa_probe(Sq=Sq)
    o = flash_attn_with_kvcache_ref(q, k, v, cache_seqlens=k.shape[1], causal=True)
    want = torch.tensor([1.0, 1.0, 1.0, 2.0, 2.0, 2.0], dtype=torch.float64).view(1, 1, 6, 1).expand_as(o)
    assert torch.allclose(o, want, atol=1e-12)    # heads 0, 1, 2 -> kv head 0; heads 3, 4, 5 -> kv head 1
