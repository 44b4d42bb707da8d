"""Adapter around oracle/_ref (the REAL reference allocator compiled against a fake CUDA driver by
oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.  Available only where /root/reference was
present at build time; ``available()`` says so.  The reference keeps process-global state
(/root/reference/vattention/apis.h:1, utils.h:12-81), so one RefImpl is live at a time.
"""
from __future__ import annotations

import ctypes
import glob
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_mod = None
_lib = None


def _so_path():
    c = glob.glob(os.path.join(_HERE, "_ref", "vattention_ref*.so"))
    return c[0] if c else None


def available() -> bool:
    return _so_path() is not None


def _load():
    global _mod, _lib
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        path = _so_path()
        if path is None:
            raise RuntimeError("oracle/_ref is not built (run oracle/build_ref.sh where /root/reference exists)")
        spec = importlib.util.spec_from_file_location("vattention_ref", path)
        _mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mod)
        _lib = ctypes.CDLL(path)
        _lib.ref_dump_state.restype = ctypes.c_long
        _lib.ref_dump_pagemap.restype = ctypes.c_long
        _lib.fakecuda_log_size.restype = ctypes.c_long
        _lib.fakecuda_log_read.restype = ctypes.c_long
    return _mod, _lib


class RefImpl:
    def __init__(self, cfg: dict):
        import torch
        self.m, self.lib = _load()
        self.cfg = cfg
        self.lib.ref_reset()
        self.lib.fakecuda_reset()
        dtype = {1: torch.int8, 2: torch.float16, 4: torch.float32}[cfg["itemsize"]]
        self.tensors = self.m.init_kvcache(cfg["num_layers"], cfg["num_kv_heads"], cfg["head_size"],
                                           cfg["max_batch_size"], cfg["max_context_length"], 0, dtype,
                                           cfg["page_size"], cfg["megacache"])
        self.bases = [t.data_ptr() for t in self.tensors]
        self.virt_total = None
        self.lib.fakecuda_log_clear()
        for n in ("reserve_physical_pages", "alloc_new_batch_idx", "free_batch_idx", "step", "step_async",
                  "num_free_kvblocks", "map_common_pages", "cleanup"):
            setattr(self, n, getattr(self.m, n))

    def set_deferred_reclamation(self, v):
        self.m.set_deferred_reclamation(bool(v))

    def _tensor_of(self, ptr: int):
        for i, b in enumerate(self.bases):
            if b <= ptr < b + self.virt_total:
                return i, ptr - b
        raise AssertionError("address outside every reserved range: %x" % ptr)

    def snapshot(self, full: bool = False) -> dict:
        self.m.set_verbose(False)      # the OOM path flips verbose on (vattention.cu:283)
        B = self.cfg["max_batch_size"]
        cap = 7 + 2 * B + (1 << 20)
        buf = (ctypes.c_ulonglong * cap)()
        n = self.lib.ref_dump_state(buf, ctypes.c_long(cap))
        assert n > 0
        pool = buf[5]
        self.virt_total = buf[4]
        s = {"mapped": list(buf[7:7 + B]), "lens": list(buf[7 + B:7 + 2 * B]), "pool": int(pool)}
        if full:
            s["pool_handles"] = list(buf[7 + 2 * B:7 + 2 * B + pool])
            rows = int(buf[6])
            pm = (ctypes.c_ulonglong * (5 * max(rows, 1)))()
            got = self.lib.ref_dump_pagemap(pm, ctypes.c_long(max(rows, 1)))
            s["pagemap"] = sorted([list(pm[5 * i:5 * i + 5]) for i in range(got)])
            nlog = self.lib.fakecuda_log_size()
            lb = (ctypes.c_ulonglong * (4 * max(nlog, 1)))()
            self.lib.fakecuda_log_read(ctypes.c_long(0), ctypes.c_long(nlog), lb)
            self.lib.fakecuda_log_clear()
            ops = []
            for i in range(nlog):
                kind, a, b, c = lb[4 * i:4 * i + 4]
                if kind in (2, 8):
                    ops.append(["create", int(a)])
                elif kind in (3, 9):
                    t, off = self._tensor_of(a)
                    ops.append(["map", t, off, int(c)])
                elif kind == 4:
                    t, off = self._tensor_of(a)
                    ops.append(["access", t, off])
                elif kind == 5:
                    if b == self.virt_total:
                        continue        # whole-tensor unmap at cleanup (cudaInternal.h:86-87)
                    t, off = self._tensor_of(a)
                    ops.append(["unmap", t, off])
            s["ops"] = ops
        return s

    def tensor_info(self):
        t = self.tensors[0]
        return {"n": len(self.tensors), "shape": list(t.shape), "stride": list(t.stride()),
                "device": str(t.device), "dtype": str(t.dtype)}
