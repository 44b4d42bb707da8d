"""Call-trace generation and replay for the page-manager parity tests.  TEST INFRASTRUCTURE ONLY.

A trace is a JSON-able dict:
    {"config": {num_layers, num_kv_heads, head_size, max_batch_size, max_context_length,
                itemsize, page_size, megacache},
     "ops": [[name, *args], ...]}
with ops  reserve(bytes) | alloc(seqlen) | free(slot) | step(lens, eager) | step_async(lens)
        | nfree() | set_deferred(flag) | map_common(ntokens) | cleanup()

``replay(impl, ops)`` drives any implementation exposing the reference's module-level API
(/root/reference/vattention/apis.h:1-63) plus ``snapshot()`` and returns one record per op:
    {"ret": value | None, "err": message | None, "mapped": [...], "lens": [...], "pool": n}
The workload generators mirror how sarathi-lean drives the allocator
(/root/reference/sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py:91-157):
alloc_new_batch_idx for every new sequence, one step/step_async per iteration with the full
curr_seq_lens vector, free_batch_idx on completion, num_free_kvblocks at iteration start.
"""
from __future__ import annotations

import random
from typing import Callable, List


def gen_serving_trace(cfg: dict, seed: int, iters: int, pool_groups: int, use_async: bool = True,
                      max_new_per_iter: int = 2, chunk: int = 0, p_finish: float = 0.02,
                      disable_deferred: bool = False, admission: bool = True) -> dict:
    """Engine-shaped workload: admissions, (chunked) prefill, decode growth, completions.
    With ``admission`` the engine only admits a sequence when num_free_kvblocks() covers it and
    the remaining growth of the running ones (count-based admission as in
    /root/reference/sarathi-lean/sarathi/core/block_space_manager/vattention_block_space_manager.py:36-66)."""
    rng = random.Random(seed)
    B, ctx = cfg["max_batch_size"], cfg["max_context_length"]
    L = cfg["num_layers"]
    ops: List[list] = []
    ops.append(["reserve", pool_groups * 2 * L * cfg["page_size"] + rng.randrange(cfg["page_size"])])
    if disable_deferred:
        ops.append(["set_deferred", False])
    lens = [0] * B            # engine-side curr_seq_lens
    target = {}               # slot -> (prompt_len, total_len)
    for _ in range(iters):
        ops.append(["nfree"])
        # admissions (engine allocates the slot with the *first chunk's* context length)
        for _ in range(rng.randrange(max_new_per_iter + 1)):
            if len(target) >= B:
                break
            prompt = rng.randrange(1, max(2, ctx - 8))
            total = min(ctx, prompt + rng.randrange(1, 64))
            first = min(prompt, chunk) if chunk else prompt
            ops.append(["alloc", first, {"prompt": prompt, "total": total}])
        ops.append(["_commit_allocs"])      # resolved at replay time (slot ids come from the impl)
        ops.append(["_step", "async" if use_async else "sync", chunk])
        if rng.random() < 0.5:
            ops.append(["nfree"])
        ops.append(["_finish", p_finish, rng.random()])
    ops.append(["cleanup"])
    return {"config": cfg, "ops": ops, "kind": "serving", "seed": seed, "admission": admission}


def gen_adversarial_trace(cfg: dict, seed: int, n_ops: int, pool_groups: int) -> dict:
    """Arbitrary API calls: lengths jump, shrink, slots freed mid-flight, pool exhaustion."""
    rng = random.Random(seed)
    B, ctx = cfg["max_batch_size"], cfg["max_context_length"]
    L = cfg["num_layers"]
    ops: List[list] = [["reserve", pool_groups * 2 * L * cfg["page_size"]]]
    lens = [0] * B
    for _ in range(n_ops):
        x = rng.random()
        if x < 0.35:
            for r in range(B):
                y = rng.random()
                if y < 0.3:
                    lens[r] = 0
                elif y < 0.6:
                    lens[r] = min(ctx, lens[r] + rng.randrange(0, 3))
                elif y < 0.8:
                    lens[r] = rng.randrange(0, ctx + 1)
            ops.append(["step_async", list(lens)])
        elif x < 0.55:
            eager = rng.random() < 0.5
            for r in range(B):
                if rng.random() < 0.4:
                    lens[r] = rng.randrange(0, ctx + 1) if rng.random() < 0.5 else 0
            ops.append(["step", list(lens), eager])
        elif x < 0.70:
            ops.append(["alloc", rng.randrange(1, ctx + 1)])
        elif x < 0.80:
            r = rng.randrange(B)
            lens[r] = 0
            ops.append(["free", r])
        elif x < 0.92:
            ops.append(["nfree"])
        elif x < 0.96:
            ops.append(["set_deferred", rng.random() < 0.5])
        else:
            ops.append(["reserve", rng.randrange(1, 4) * 2 * L * cfg["page_size"] * pool_groups])
    ops.append(["cleanup"])
    return {"config": cfg, "ops": ops, "kind": "adversarial", "seed": seed}


def resolve(trace: dict, impl_factory: Callable[[dict], object]) -> List[list]:
    """Turn the engine-shaped pseudo-ops (_commit_allocs/_step/_finish) into concrete API calls
    by running them once against an implementation (slot ids are decided by the allocator)."""
    impl = impl_factory(trace["config"])
    cfg = trace["config"]
    B = cfg["max_batch_size"]
    lens = [0] * B
    seqs = {}      # slot -> dict(prompt,total,done)
    pending = []
    out: List[list] = []
    per_tok = cfg["num_kv_heads"] * cfg["head_size"] * cfg["itemsize"] * (cfg["num_layers"] if cfg["megacache"] else 1)
    tpp = cfg["page_size"] // per_tok
    pages = lambda n: (n + tpp - 1) // tpp
    for op in trace["ops"]:
        name = op[0]
        if name == "alloc" and len(op) == 3:
            if trace.get("admission", False):
                free_now = impl.num_free_kvblocks()
                out.append(["nfree"])
                if free_now >= (1 << 63):
                    free_now -= 1 << 64
                growth = sum(pages(s["total"]) - pages(s["done"]) for s in seqs.values())
                growth += sum(pages(m["total"]) for _, _, m in pending)
                if free_now - growth - pages(op[2]["total"]) < 1:
                    continue
            slot = impl.alloc_new_batch_idx(op[1])
            out.append(["alloc", op[1]])
            if slot >= 0:
                pending.append((slot, op[1], op[2]))
        elif name == "_commit_allocs":
            for slot, first, meta in pending:
                seqs[slot] = {"prompt": meta["prompt"], "total": meta["total"], "done": first}
                lens[slot] = first
            pending = []
        elif name == "_step":
            mode, chunk = op[1], op[2]
            call = ["step_async", list(lens)] if mode == "async" else ["step", list(lens), True]
            out.append(call)
            try:
                if mode == "async":
                    impl.step_async(list(lens))
                else:
                    impl.step(list(lens), True)
            except RuntimeError:
                pass
            # advance engine-side lengths for the next iteration
            for slot, s in seqs.items():
                if s["done"] < s["prompt"]:
                    s["done"] = min(s["prompt"], s["done"] + (chunk or s["prompt"]))
                else:
                    s["done"] = min(s["total"], s["done"] + 1)
                lens[slot] = s["done"]
        elif name == "_finish":
            p, _ = op[1], op[2]
            rng = random.Random(int(op[2] * 1e9))
            for slot in sorted(seqs):
                s = seqs[slot]
                if s["done"] >= s["total"] or rng.random() < p:
                    out.append(["free", slot])
                    impl.free_batch_idx(slot)
                    lens[slot] = 0
                    del seqs[slot]
        else:
            out.append(op)
            try:
                _apply(impl, op)
            except RuntimeError:
                pass
    return out


def _apply(impl, op):
    name = op[0]
    if name == "reserve":
        return impl.reserve_physical_pages(op[1])
    if name == "alloc":
        return impl.alloc_new_batch_idx(op[1])
    if name == "free":
        return impl.free_batch_idx(op[1])
    if name == "step":
        return impl.step(list(op[1]), bool(op[2]))
    if name == "step_async":
        return impl.step_async(list(op[1]))
    if name == "nfree":
        return impl.num_free_kvblocks()
    if name == "set_deferred":
        return impl.set_deferred_reclamation(bool(op[1]))
    if name == "map_common":
        return impl.map_common_pages(op[1])
    if name == "cleanup":
        return impl.cleanup()
    raise ValueError(name)


def replay(impl, ops: List[list], full: bool = False) -> List[dict]:
    """Apply concrete ops; record return value / error text and the observable state after each."""
    recs = []
    for op in ops:
        ret, err = None, None
        try:
            ret = _apply(impl, op)
        except RuntimeError as e:           # pybind maps std::runtime_error -> RuntimeError
            err = str(e)
        except ValueError as e:             # explicit argument errors (reference: assert / undefined behaviour)
            err = "ValueError: " + str(e)
        snap = impl.snapshot(full)
        rec = {"ret": ret, "err": err}
        rec.update(snap)
        recs.append(rec)
    return recs


class OracleImpl:
    """Adapter: oracle.pagemgr.PageManagerOracle behind the trace interface."""

    def __init__(self, cfg: dict, shared_page_refcount: bool = False):
        from oracle.pagemgr import PageManagerOracle
        self.o = PageManagerOracle(cfg["num_layers"], cfg["num_kv_heads"], cfg["head_size"],
                                   cfg["max_batch_size"], cfg["max_context_length"], cfg["itemsize"],
                                   cfg["page_size"], cfg["megacache"], shared_page_refcount=shared_page_refcount)
        for n in ("reserve_physical_pages", "alloc_new_batch_idx", "free_batch_idx", "step", "step_async",
                  "num_free_kvblocks", "set_deferred_reclamation", "map_common_pages", "cleanup"):
            setattr(self, n, getattr(self.o, n))

    def snapshot(self, full: bool = False) -> dict:
        s = {"mapped": list(self.o.mapped_pages), "lens": list(self.o.curr_seq_lengths), "pool": len(self.o.pool)}
        if full:
            s["pool_handles"] = list(self.o.pool)
            s["pagemap"] = [list(t) for t in self.o.state()["pagemap"]]
            s["ops"] = [list(t) for t in self.o.ops]
            self.o.ops.clear()
        return s


def truncate_for_reference(ops: List[list], oracle_recs: List[dict]) -> List[list]:
    """The reference's step_async throws its OOM error with the GIL released
    (/root/reference/vattention/apis.h:31-35), which crashes the interpreter; a trace that is to
    be replayed on the real reference therefore stops before the first step_async that OOMs."""
    for i, (op, rec) in enumerate(zip(ops, oracle_recs)):
        if op[0] == "step_async" and rec["err"] is not None:
            return ops[:i] + [["cleanup"]]
    return ops


def normalize_ops(recs: List[dict], page_size: int) -> List[dict]:
    """The reference's small-page backend (page_size != 2 MiB, uvmInternal.h:237-247, mux.h:60-65)
    issues no set-access call and *no unmap call at all* (pages just return to the pool); the
    oracle logs backend-neutral ops, so for those configurations compare creates and maps only."""
    if page_size == 2 * 1024 * 1024:
        return recs
    out = []
    for r in recs:
        r = dict(r)
        if "ops" in r:
            r["ops"] = [o for o in r["ops"] if o[0] in ("create", "map")]
        out.append(r)
    return out
