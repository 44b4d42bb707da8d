#!/usr/bin/env python3
"""Golden vectors for the attention-wrapper boundary, produced by IMPORTING THE REFERENCE (this container only):
the reference's own VAttentionFlashAttentionWrapper.forward
(/root/reference/sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py:110-224) runs on CPU tensors
with `flash_attn_with_kvcache` / `cache_flat` bound to the CPU oracle (oracle/attn.py) — so the fixture pins the reference's
DATAFLOW (token order, slot selection, cache_flat offsets, cache_seqlens, the [:, :max_cache_len] decode view, in-kernel append)
on top of the restated arithmetic.  The GPU test (tests/test_gpu_wrapper_golden.py) replays the same schedule through this
package's wrapper + cache engine + kernels and compares outputs and final cache contents.

Writes tests/golden/wrapper_hybrid_trace.npz: the reference's outputs per iteration (fp16) + SHA-256 of the seeded inputs and of
the final K / V cache contents (cache_flat and the in-kernel append are bit-exact, so a digest suffices).
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.attn import cache_flat_ref, flash_attn_with_kvcache_ref  # noqa: E402
from tests import ref_loader  # noqa: E402
from tests.wrapper_schedule import HQ, HKV, D, MAX_BATCH, MAX_CTX, make_inputs, schedule  # noqa: E402


def fa_cpu(q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, cache_batch_idx=None, block_table=None, softmax_scale=None, causal=False):
    assert block_table is None
    return flash_attn_with_kvcache_ref(q, k_cache, v_cache, k, v, cache_seqlens=cache_seqlens, cache_batch_idx=cache_batch_idx,
                                       softmax_scale=softmax_scale, causal=causal, math="f32")


def main():
    if ref_loader.available() != "source":
        raise SystemExit("needs /root/reference (this container)")
    from vattention_amd.replay import ModelConfig, ParallelConfig
    model = ModelConfig(name="tiny", num_layers=1, num_q_heads=HQ, num_kv_heads=HKV, head_size=D, dtype=torch.float16, max_model_len=MAX_CTX)
    outs, h = [], hashlib.sha256()
    with ref_loader.loaded(cpu_kernels=(fa_cpu, cache_flat_ref)) as ref:
        w = ref.wrapper
        w.init(model, ParallelConfig(1, 1), 0, torch.device("cpu"))
        kc = torch.zeros(MAX_BATCH, MAX_CTX, HKV, D, dtype=torch.float16)
        vc = torch.zeros(MAX_BATCH, MAX_CTX, HKV, D, dtype=torch.float16)
        for it, (mds, slots_p, slots_d) in enumerate(schedule()):
            q, k, v = make_inputs(it, mds)
            for t in (q, k, v):
                h.update(t.numpy().tobytes())
            w.begin_forward(mds)
            w.set_batch_idx(torch.tensor(slots_p + slots_d, dtype=torch.int32), torch.tensor(slots_d, dtype=torch.int32))
            out = w.forward(q, k, v, (kc, vc), D ** -0.5, 0)
            w.end_forward()
            outs.append(out.numpy().copy())
    path = os.path.join(ROOT, "tests", "golden", "wrapper_hybrid_trace.npz")
    hk = hashlib.sha256(kc.numpy().tobytes()).digest()
    hv = hashlib.sha256(vc.numpy().tobytes()).digest()
    np.savez_compressed(path, inputs_sha256=np.frombuffer(h.digest(), dtype=np.uint8), k_cache_sha256=np.frombuffer(hk, dtype=np.uint8),
                        v_cache_sha256=np.frombuffer(hv, dtype=np.uint8), **{"out_%d" % i: o for i, o in enumerate(outs)})
    print(path, os.path.getsize(path) // 1024, "KiB", len(outs), "iterations")


if __name__ == "__main__":
    main()
