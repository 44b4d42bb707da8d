"""Adapters that put the product's C++ page manager behind the trace-replay interface (oracle/trace.py)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_SO = os.path.join(HERE, "native", "libvattn_fake_backend.so")
_fake = None


def fake():
    global _fake
    if _fake is None:
        if not os.path.exists(FAKE_SO):
            raise RuntimeError("tests/native/libvattn_fake_backend.so missing: run __graft_entry__.build()")
        _fake = C.CDLL(FAKE_SO)
        _fake.vattn_fake_backend_ops.restype = C.c_void_p
        _fake.vattn_fake_reset.argtypes = [C.c_uint64, C.c_uint64]
        _fake.vattn_fake_mapped.restype = C.c_int64
        _fake.vattn_fake_mapped.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        _fake.vattn_fake_fail_create_after.argtypes = [C.c_uint64]
        _fake.vattn_fake_fail_map_after.argtypes = [C.c_uint64]
        _fake.vattn_fake_quiesce_count.restype = C.c_uint64
        _fake.vattn_fake_fence_wait_count.restype = C.c_uint64
    return _fake


def fake_counters():
    buf = (C.c_uint64 * 12)()
    fake().vattn_fake_counters(buf)
    names = ["violations", "n_create", "n_map", "n_access", "n_unmap", "n_release", "live_handles", "mapped_pages",
             "accessible_pages", "reserved_ranges", "n_flush", "stale_vas"]
    return dict(zip(names, [int(x) for x in buf]))


def fake_mapped():
    cap = 1 << 16
    while True:
        buf = (C.c_uint64 * (4 * cap))()
        n = fake().vattn_fake_mapped(buf, cap)
        if n >= 0:
            return [tuple(int(x) for x in buf[4 * i:4 * i + 4]) for i in range(n)]
        cap = -n


class ProductImpl:
    """libvattn_amd.so's PageManager on the fake backend.  Page ids are shifted by 1000 so that they
    compare equal to the handle numbers of the reference run (fake CUDA driver numbers from 1000)."""
    HANDLE_BASE = 1000

    def __init__(self, cfg, flags=0, min_gran=4096):
        from vattention_amd.page_manager import PageManager
        f = fake()
        f.vattn_fake_reset(min_gran, 2 << 20)
        self.pm = PageManager(cfg["num_layers"], cfg["num_kv_heads"], cfg["head_size"], cfg["max_batch_size"],
                              cfg["max_context_length"], cfg["itemsize"], 0, cfg["page_size"], cfg["megacache"],
                              flags=flags, backend=f.vattn_fake_backend_ops())
        for n in ("reserve_physical_pages", "alloc_new_batch_idx", "free_batch_idx", "step", "step_async",
                  "num_free_kvblocks", "set_deferred_reclamation", "map_common_pages", "cleanup"):
            setattr(self, n, getattr(self.pm, n))

    def snapshot(self, full=False):
        self.pm.set_verbose(False)
        st = self.pm.state()
        s = {"mapped": st["mapped"], "lens": st["lens"], "pool": st["pool"]}
        if full:
            s["pool_handles"] = [p + self.HANDLE_BASE for p in st["pool_ids"]]
            s["pagemap"] = sorted([r[0], r[1], r[2], r[3] + self.HANDLE_BASE, r[4] + self.HANDLE_BASE] for r in self.pm.pagemap())
        return s

    def mapped_ranges(self):
        """(tensor, offset) of every page the backend currently has mapped AND accessible."""
        self.pm.wait()
        bases = [self.pm.tensor_base(i) for i in range(self.pm.num_tensors)]
        total = int(self.pm.layout.virt_bytes_total)
        out = set()
        for va, nbytes, _h, acc in fake_mapped():
            assert acc == 1, "mapped page without access rights"
            assert nbytes == self.pm.layout.page_size
            for i, b in enumerate(bases):
                if b <= va < b + total:
                    out.add((i, va - b))
                    break
            else:
                raise AssertionError("mapping outside reserved ranges")
        return out
