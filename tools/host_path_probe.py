#!/usr/bin/env python3
"""Host cost of one decode layer-call (the Python path between two launches): one short sequence (the GPU work is a few microseconds, so
the loop is host-bound by construction), L layers, op timers off and on.  A batch-1 decode at 128 k context is ~50 us of GPU time per
layer: whatever the host needs beyond that shows up as an idle GPU in bench.py's tensor-parallel lines.
usage: python tools/host_path_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vattention_amd.attention.timers import drain_op_timers_detail, enable_op_timers  # noqa: E402
from vattention_amd.replay import CacheConfig, HotPathRunner, ModelConfig, ParallelConfig, Sequence, SequenceMetadata  # noqa: E402


def main():
    torch.zeros(1, device="cuda:0")
    model = ModelConfig.named("yi-34b", dtype=torch.float16, max_model_len=4096, attention_backend="fa_vattn")
    r = HotPathRunner(model, ParallelConfig(2, 1), CacheConfig(page_size=2 << 20, max_batch_size=4, memory_for_gpu=8 << 30, vattn_keep_layout=True))
    try:
        s = Sequence(0, 200, 4000)
        r.sample_kv_util = False
        r.run_iteration([SequenceMetadata(s, 200, True)])
        for timers in (False, True):
            enable_op_timers(timers, every=1)
            for _ in range(5):
                r.run_iteration([SequenceMetadata(s, 0, False)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 50
            for _ in range(n):
                r.run_iteration([SequenceMetadata(s, 0, False)])
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            drain_op_timers_detail()
            print("op timers %-3s: %.1f us of host time per layer-call (%d layers, %d iterations; %.1f us incl. the GPU drain)" % (
                "on" if timers else "off", t_host / n / r.L * 1e6, r.L, n, t_all / n / r.L * 1e6))
        enable_op_timers(False)
    finally:
        r.close()


if __name__ == "__main__":
    main()
