"""This package's fa_vattn wrapper + kernels on the GPU against golden vectors produced by the REFERENCE's own wrapper
(oracle/gen_golden_wrapper.py: the reference's VAttentionFlashAttentionWrapper.forward run on CPU over the oracle kernels, in the
container that has /root/reference).  Same schedule (chunked prefill, whole prompt, hybrid iterations, decode batches), same seeded
inputs, same slots: outputs within the fp16 tolerance, final K/V cache contents bit-identical."""
import hashlib
import os

import numpy as np
import pytest
import torch

from tests.wrapper_schedule import HQ, HKV, D, MAX_BATCH, MAX_CTX, make_inputs, schedule

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrapper_hybrid_trace.npz")


@pytest.mark.parametrize("backend", ["fa_vattn", "fa_pod"])
def test_wrapper_matches_the_reference_wrappers_golden_outputs(backend):
    from vattention_amd.attention import get_attention_wrapper, set_attention_backend
    from vattention_amd.replay import ModelConfig, ParallelConfig
    g = np.load(GOLD)
    dev = torch.device("cuda:0")
    model = ModelConfig(name="tiny", num_layers=1, num_q_heads=HQ, num_kv_heads=HKV, head_size=D, dtype=torch.float16, max_model_len=MAX_CTX,
                        attention_backend=backend)
    set_attention_backend(backend)
    w = get_attention_wrapper()
    w.init(model, ParallelConfig(1, 1), 0, dev)
    kc = torch.zeros(MAX_BATCH, MAX_CTX, HKV, D, dtype=torch.float16, device=dev)
    vc = torch.zeros(MAX_BATCH, MAX_CTX, HKV, D, dtype=torch.float16, device=dev)
    h = hashlib.sha256()
    for it, (mds, slots_p, slots_d) in enumerate(schedule()):
        q, k, v = make_inputs(it, mds)
        for t in (q, k, v):
            h.update(t.numpy().tobytes())
        w.begin_forward(mds)
        w.set_batch_idx(torch.tensor(slots_p + slots_d, dtype=torch.int32, device=dev), torch.tensor(slots_d, dtype=torch.int32, device=dev))
        out = w.forward(q.to(dev), k.to(dev), v.to(dev), (kc, vc), D ** -0.5, 0)
        w.end_forward()
        torch.cuda.synchronize()
        want = torch.from_numpy(g["out_%d" % it]).double()
        err = (out.double().cpu() - want).abs()
        assert bool((err <= 3e-3 + 3e-3 * want.abs()).all()), "iteration %d: max err %.3e vs the reference wrapper's output" % (it, err.max().item())
    assert h.digest() == bytes(g["inputs_sha256"]), "seeded inputs differ from the ones the golden run saw"
    assert hashlib.sha256(kc.cpu().numpy().tobytes()).digest() == bytes(g["k_cache_sha256"])
    assert hashlib.sha256(vc.cpu().numpy().tobytes()).digest() == bytes(g["v_cache_sha256"])
