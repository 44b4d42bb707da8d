"""Bookkeeping CPU baseline: allocator steps per second with FREE driver calls — the reference's own allocator
(oracle/_ref: vattention.cu compiled here against a fake CUDA driver, call log off), this package's C++ page manager on the fake
physical backend (inline and with its mapper thread), and the Python oracle, all replaying the SAME concrete call sequence of an
engine-shaped trace (admissions, prefill, decode growth, completions; oracle/trace.py) at BASELINE.json's allocator geometries.
No GPU.  What it prices is the host arithmetic + containers per step_async; on hardware the driver calls dominate.

    python tools/pagemgr_steps_bench.py [--iters 400] > profiles/rNN_pagemgr_steps.txt
"""
import os
import sys
import time
import functools
print = functools.partial(print, flush=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_adapter, trace as T  # noqa: E402
from tests.impls import ProductImpl, fake  # noqa: E402

GEOMS = {
    "configs[1] yi-6b TP1, 2 MiB pages":            dict(num_layers=32, num_kv_heads=4, head_size=128, max_batch_size=16, max_context_length=32768, itemsize=2, page_size=2 << 20, megacache=False),
    "configs[2] llama-3-8b TP1, 64 KiB pages":      dict(num_layers=32, num_kv_heads=8, head_size=128, max_batch_size=256, max_context_length=32768, itemsize=2, page_size=64 << 10, megacache=False),
    "configs[4] llama-3-70b TP8 rank, 256 KiB":     dict(num_layers=80, num_kv_heads=1, head_size=128, max_batch_size=256, max_context_length=32768, itemsize=2, page_size=256 << 10, megacache=False),
    "llama-3-70b TP8 rank, megacache 8 MiB pages":  dict(num_layers=80, num_kv_heads=1, head_size=128, max_batch_size=256, max_context_length=32768, itemsize=2, page_size=8 << 20, megacache=True),
}


def pool_groups(cfg):
    group = cfg["page_size"] * (2 if cfg["megacache"] else 2 * cfg["num_layers"])
    return int(288e9 * 0.9) // group


def run(impl, ops):
    """Apply the concrete ops; returns (seconds, seconds inside step calls, step calls, errors)."""
    t_step = 0.0
    n_step = errs = 0
    t0 = time.perf_counter()
    for op in ops:
        name = op[0]
        try:
            if name in ("step_async", "step"):
                a = time.perf_counter()
                impl.step_async(op[1]) if name == "step_async" else impl.step(op[1], op[2])
                t_step += time.perf_counter() - a
                n_step += 1
            else:
                T._apply(impl, op)
        except RuntimeError:
            errs += 1
    return time.perf_counter() - t0, t_step, n_step, errs


def measure(geom="configs[1] yi-6b TP1, 2 MiB pages", iters=400):
    """One geometry, as a dict (bench.py's cpu_baseline.bookkeeping): microseconds inside a step_async call for the reference's
    allocator (oracle/_ref, None when it is not built) and for this package's manager, same concrete call sequence, driver calls free."""
    cfg = GEOMS[geom]
    tr = T.gen_serving_trace(cfg, 7, iters=iters, pool_groups=pool_groups(cfg), use_async=True, max_new_per_iter=2, chunk=0, p_finish=0.01)
    ops = T.resolve(tr, T.OracleImpl)
    recs = T.replay(T.OracleImpl(cfg), ops)
    ops = T.truncate_for_reference(ops, recs)
    out = {"geometry": geom, "engine_iterations": iters, "ops": len(ops), "step_async_calls": sum(1 for o in ops if o[0] in ("step", "step_async")),
           "reference_us_per_step_async": None}
    if ref_adapter.available():
        r = ref_adapter.RefImpl(cfg)
        r.lib.fakecuda_enable_log(0)
        sec, t_step, n, _ = run(r, ops)
        r.lib.fakecuda_enable_log(1)
        out["reference_us_per_step_async"] = round(1e6 * t_step / max(n, 1), 2)
    for flags, key in ((4, "this_manager_inline_us_per_step_async"), (0, "this_manager_mapper_thread_us_per_step_async")):
        p = ProductImpl(cfg, flags=flags)
        fake().vattn_fake_set_validate(0)
        sec, t_step, n, _ = run(p, ops)
        p.pm.close()
        out[key] = round(1e6 * t_step / max(n, 1), 2)
    fake().vattn_fake_set_validate(1)
    return out


def main():
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 400
    print("# allocator steps/s with free driver calls, %d engine iterations per trace, %d host cores" % (iters, os.cpu_count()))
    for name, cfg in GEOMS.items():
        tr = T.gen_serving_trace(cfg, 7, iters=iters, pool_groups=pool_groups(cfg), use_async=True, max_new_per_iter=2, chunk=0, p_finish=0.01)
        ops = T.resolve(tr, T.OracleImpl)
        recs = T.replay(T.OracleImpl(cfg), ops)
        ops = T.truncate_for_reference(ops, recs)
        steps = sum(1 for o in ops if o[0] in ("step", "step_async"))
        live = max((sum(1 for x in o[1] if x) for o in ops if o[0] == "step_async"), default=0)
        print("== %s: %d ops, %d step_async calls, up to %d live sequences, pool %d page-groups" % (name, len(ops), steps, live, pool_groups(cfg)))
        rows = []
        # (the reference asserts on page sizes other than 64 / 128 / 256 KiB and 2 MiB: uvmInternal.h:222)
        if ref_adapter.available() and cfg["page_size"] in (64 << 10, 128 << 10, 256 << 10, 2 << 20):
            r = ref_adapter.RefImpl(cfg)
            r.lib.fakecuda_enable_log(0)
            rows.append(("reference allocator (oracle/_ref, fake driver)", run(r, ops)))
            r.lib.fakecuda_enable_log(1)
        for flags, label in ((4, "this package, inline execution"), (0, "this package, mapper thread")):
            p = ProductImpl(cfg, flags=flags)
            fake().vattn_fake_set_validate(0)       # count calls only, like the reference's fake driver with its log off
            rows.append((label + " (fake backend)", run(p, ops)))
            p.pm.close()
        o = T.OracleImpl(cfg)
        rows.append(("Python oracle (oracle/pagemgr.py)", run(o, ops)))
        for label, (sec, t_step, n, errs) in rows:
            print("  %-52s %8.3f s total  %9.0f steps/s  %8.1f us inside a step call%s" % (label, sec, n / sec, 1e6 * t_step / max(n, 1), "  (%d errors)" % errs if errs else ""))


if __name__ == "__main__":
    main()
