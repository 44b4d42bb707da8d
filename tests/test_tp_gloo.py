"""N>1 path on CPU: two processes (gloo), one page manager each (C++ core, fake backend), KV sharded by head.
Asserts (a) per-rank shapes follow the reference's head division, (b) both ranks take IDENTICAL page decisions from the
same seq_lens although they never talk on the data path, (c) the control-plane min(free_blocks) reduction."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import trace as T
        from tests.impls import ProductImpl, fake_counters
        from vattention_amd.tp import heads_for_rank, min_free_kvblocks
        # world 2: Yi-34B, 56 q / 8 kv heads -> 28 / 4 per rank; world 8: Llama-3-70B, 64 / 8 -> 8 / 1 per rank
        if world == 2:
            hq, hkv = heads_for_rank(56, 8, world)
            assert (hq, hkv) == (28, 4)
        else:
            hq, hkv = heads_for_rank(64, 8, world)
            assert (hq, hkv) == (8, 1)
        cfg = dict(num_layers=3, num_kv_heads=hkv, head_size=128, max_batch_size=8, max_context_length=8192,
                   itemsize=2, page_size=256 << 10, megacache=False)
        # identical budgets on every rank, as the engine guarantees (memory_for_gpu = min over workers,
        # base_llm_engine.py:242-264) -> identical page decisions with no communication
        impl = ProductImpl(cfg, flags=0)
        tr = T.gen_serving_trace(cfg, 777, iters=40, pool_groups=60, use_async=True, chunk=1024, p_finish=0.05)
        ops = T.resolve(tr, T.OracleImpl)            # deterministic: same ops on both ranks
        # what bench.py's TP mode does inside its timed loop, every iteration: all-reduce MIN of num_free_kvblocks
        # (base_llm_engine.py:381-390) and an all-gather of a fingerprint of the page-manager state
        ok = True
        for op in ops[:60]:
            T.replay(impl, [op])
            c = impl.pm.counts()
            fp = (c["mapped_groups"] * 1000003 + c["needed_groups"]) * 1000003 + c["pool_pages"] * 31 + c["active_slots"]
            free = impl.num_free_kvblocks()
            free = free - (1 << 64) if free >= (1 << 63) else free
            assert min_free_kvblocks(free) == free                      # identical on every rank
            g = torch.empty(world, dtype=torch.int64)
            dist.all_gather_into_tensor(g, torch.tensor([fp], dtype=torch.int64))
            ok = ok and bool((g == g[0]).all())
        assert ok, "page-manager fingerprints diverged across ranks"
        impl.cleanup()
        impl = ProductImpl(cfg, flags=0)
        recs = T.replay(impl, ops)
        mapped = torch.tensor([r["mapped"] for r in recs if True], dtype=torch.int64)
        lens = torch.tensor([r["lens"] for r in recs], dtype=torch.int64)
        gathered = [torch.zeros_like(mapped) for _ in range(world)]
        dist.all_gather(gathered, mapped)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks diverged in page decisions"
        gl = [torch.zeros_like(lens) for _ in range(world)]
        dist.all_gather(gl, lens)
        assert all(torch.equal(g, gl[0]) for g in gl)
        # control plane: scheduler admits against the minimum over ranks
        impl2 = ProductImpl(cfg, flags=4)
        impl2.reserve_physical_pages((100 - 10 * rank) * 2 * cfg["num_layers"] * cfg["page_size"])
        local = impl2.num_free_kvblocks()
        assert local == 100 - 10 * rank
        assert min_free_kvblocks(local) == 100 - 10 * (world - 1)
        assert fake_counters()["violations"] == 0
        q.put((rank, "ok"))
    except Exception as e:      # surface the failure to the parent
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_head_sharded_managers_agree_across_ranks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
