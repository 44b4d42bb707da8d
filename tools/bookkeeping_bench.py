#!/usr/bin/env python3
"""CPU-only: allocator bookkeeping throughput on one engine-shaped trace (SURVEY §8d "bookkeeping CPU baseline") -
the Python oracle, the real reference allocator (oracle/_ref, fake CUDA driver) and the product's C++ page manager on
the fake backend (inline and mapper-thread mode).  Calls/s of the public API, driver calls excluded from none of them
(all three only record them).  usage: python tools/bookkeeping_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import trace as T  # noqa: E402


def apply_all(impl, ops):
    n = 0
    for op in ops:
        try:
            T._apply(impl, op)
        except (RuntimeError, ValueError):
            pass
        n += 1
    return n


def main():
    cfg = dict(num_layers=32, num_kv_heads=8, head_size=128, max_batch_size=256, max_context_length=32768, itemsize=2,
               page_size=2 << 20, megacache=False)
    tr = T.gen_serving_trace(cfg, seed=3, iters=1500, pool_groups=1800, use_async=True, max_new_per_iter=4)
    ops = T.resolve(tr, lambda c: T.OracleImpl(c))
    print("trace: %d API calls (Llama-3-8B shape, 256 slots, 32k ctx, 2 MiB pages, pool 1800 groups)" % len(ops))
    rows = []
    t0 = time.perf_counter(); n = apply_all(T.OracleImpl(cfg), ops); rows.append(("oracle (Python restatement)", n / (time.perf_counter() - t0)))
    try:
        from oracle.ref_adapter import RefImpl
        recs = T.replay(T.OracleImpl(cfg), ops)
        rops = T.truncate_for_reference(ops, recs)       # the reference crashes on an OOM inside step_async: stop before it
        impl = RefImpl(cfg)
        t0 = time.perf_counter(); n = apply_all(impl, rops)
        rows.append(("reference vattention.cu (oracle/_ref, fake driver; %d calls)" % n, n / (time.perf_counter() - t0)))
    except Exception as e:      # oracle/_ref only exists where /root/reference was present at build time
        rows.append(("reference (oracle/_ref unavailable: %s)" % type(e).__name__, float("nan")))
    from impls import ProductImpl
    for flags, name in ((4, "product C++ manager, inline execution"), (0, "product C++ manager, mapper thread")):
        impl = ProductImpl(cfg, flags=flags)
        t0 = time.perf_counter(); n = apply_all(impl, ops); dt = time.perf_counter() - t0
        impl.cleanup()
        rows.append((name, n / dt))
    for name, v in rows:
        print("  %-52s %12.0f calls/s" % (name, v))


if __name__ == "__main__":
    main()
