"""Object wrapper over the page-manager C ABI (include/vattn.h).

`PageManager(...)` with ``backend=None`` is the product: HIP VMM on the given device.  Tests pass
the address of a `vattn_backend_ops` table (tests/native/libvattn_fake_backend.so) to drive the
same C++ bookkeeping core on a CPU-only box.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from . import _lib as L


class PageManager:
    def __init__(self, num_layers: int, num_kv_heads: int, head_size: int, max_batch_size: int,
                 max_context_length: int, itemsize: int, device: int, page_size: int, megacache: bool,
                 flags: int = 0, backend: Optional[int] = None):
        self._lib = L.lib()
        self.cfg = L.VattnConfig(num_layers, num_kv_heads, head_size, max_batch_size, max_context_length,
                                 itemsize, device, page_size, 1 if megacache else 0, flags)
        self._h = C.c_void_p()
        rc = self._lib.vattn_create(C.byref(self.cfg), C.c_void_p(backend) if backend else None, C.byref(self._h))
        if rc != L.VATTN_OK:
            msg = self._lib.vattn_last_error(self._h).decode() if self._h else "HIP VMM backend unavailable (no device?)"
            if self._h:
                self._lib.vattn_destroy(self._h)
                self._h = C.c_void_p()
            raise (ValueError if rc == L.VATTN_ERR_INVALID else RuntimeError)(msg)
        self.max_batch_size = max_batch_size
        lay = L.VattnLayout()
        self._lib.vattn_get_layout(self._h, C.byref(lay))
        self.layout = lay
        self._lens_t = C.c_uint64 * max_batch_size

    # -- errors: reproduce the reference's exception types/messages (SURVEY §8b "Errors") --
    def _check(self, rc: int):
        if rc == L.VATTN_OK:
            return
        msg = self._lib.vattn_last_error(self._h).decode()
        if rc == L.VATTN_ERR_INVALID:
            raise ValueError(msg)
        raise RuntimeError(msg)          # std::runtime_error -> RuntimeError in the reference's pybind layer

    @property
    def num_tensors(self) -> int:
        return self._lib.vattn_num_tensors(self._h)

    def tensor_base(self, i: int) -> int:
        return self._lib.vattn_tensor_base(self._h, i)

    def shape(self) -> List[int]:
        return [int(self.layout.shape[i]) for i in range(self.layout.ndim)]

    def stride(self) -> List[int]:
        return [int(self.layout.stride[i]) for i in range(self.layout.ndim)]

    def reserve_physical_pages(self, free_memory: int) -> int:
        n = self._lib.vattn_reserve_physical_pages(self._h, int(free_memory))
        if n < 0 and n >= -4:
            self._check(int(n))
        return int(n)

    def _lens(self, seq_lens: Sequence[int]):
        if len(seq_lens) != self.max_batch_size:
            raise ValueError("seq_lens must have max_batch_size entries")
        return self._lens_t(*[int(x) for x in seq_lens])

    def step(self, seq_lens: Sequence[int], eager_reclaim: bool) -> None:
        self._check(self._lib.vattn_step(self._h, self._lens(seq_lens), self.max_batch_size, 1 if eager_reclaim else 0))

    def step_async(self, seq_lens: Sequence[int]) -> None:
        # ctypes drops the GIL for the duration of the call (apis.h:32-34 Py_BEGIN_ALLOW_THREADS)
        self._check(self._lib.vattn_step_async(self._h, self._lens(seq_lens), self.max_batch_size))

    def wait(self) -> None:
        self._check(self._lib.vattn_wait(self._h))

    def alloc_new_batch_idx(self, seqlen: int) -> int:
        return self._lib.vattn_alloc_new_batch_idx(self._h, int(seqlen))

    def free_batch_idx(self, slot: int, stream: Optional[int] = None) -> None:
        """`stream` (a raw hipStream_t, 0 = the default stream): also records the slot's fence there (include/vattn.h)."""
        if stream is None:
            self._check(self._lib.vattn_free_batch_idx(self._h, int(slot)))
        else:
            self._check(self._lib.vattn_free_batch_idx_on_stream(self._h, int(slot), C.c_void_p(stream)))

    def premap(self, seqlen: int) -> int:
        return self._lib.vattn_premap(self._h, int(seqlen))

    def wait_pool_ready(self, timeout_ms: int = -1) -> int:
        """Handles the mapper thread still has to create ahead of demand (0 = ready); blocks up to timeout_ms (< 0: until ready)."""
        return int(self._lib.vattn_wait_pool_ready(self._h, int(timeout_ms)))

    def cancel_premap(self, slot: int) -> None:
        self._check(self._lib.vattn_cancel_premap(self._h, int(slot)))

    def wait_layer(self, layer: int) -> None:
        if self._lib.vattn_wait_layer(self._h, int(layer)) != L.VATTN_OK:
            # (vattn_wait_layer does not touch the manager's message buffer: include/vattn.h)
            raise RuntimeError("layer-ordered mapping failed on the mapper thread; the next step()/wait() reports the driver's error")

    def layers_ready(self) -> int:
        return int(self._lib.vattn_layers_ready(self._h))

    def set_sync_layers(self, n: int) -> None:
        self._check(self._lib.vattn_set_sync_layers(self._h, int(n)))

    def num_free_kvblocks(self) -> int:
        return int(self._lib.vattn_num_free_kvblocks(self._h))

    def set_deferred_reclamation(self, on: bool) -> None:
        self._lib.vattn_set_deferred_reclamation(self._h, 1 if on else 0)

    def set_verbose(self, on: bool) -> None:
        self._lib.vattn_set_verbose(self._h, 1 if on else 0)

    def map_common_pages(self, num_tokens: int) -> None:
        self._check(self._lib.vattn_map_common_pages(self._h, int(num_tokens)))

    def show_kvcache_config(self) -> None:
        self._lib.vattn_show_kvcache_config(self._h)

    def show_allocator_state(self) -> None:
        self._lib.vattn_show_allocator_state(self._h)

    def cleanup(self) -> None:
        self._check(self._lib.vattn_cleanup(self._h))

    def close(self) -> None:
        if self._h:
            self._lib.vattn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- introspection --
    def state(self) -> dict:
        B = self.max_batch_size
        cap = 3 + 2 * B + 1024
        while True:
            buf = (C.c_uint64 * cap)()
            n = self._lib.vattn_state_dump(self._h, buf, cap)
            if n >= 0:
                break
            cap = -n
        pool = int(buf[1])
        return {"mapped": list(buf[3:3 + B]), "lens": list(buf[3 + B:3 + 2 * B]), "pool": pool,
                "pool_ids": list(buf[3 + 2 * B:3 + 2 * B + pool]), "pagemap_rows": int(buf[2])}

    def pagemap(self) -> List[List[int]]:
        rows = max(1, self.state()["pagemap_rows"])
        buf = (C.c_uint64 * (5 * rows))()
        n = self._lib.vattn_pagemap_dump(self._h, buf, rows)
        return [list(buf[5 * i:5 * i + 5]) for i in range(max(0, n))]

    def counts(self) -> dict:
        buf = (C.c_uint64 * 4)()
        self._lib.vattn_get_counts(self._h, buf)
        return {"pool_pages": int(buf[0]), "mapped_groups": int(buf[1]), "needed_groups": int(buf[2]), "active_slots": int(buf[3])}

    def stats(self) -> dict:
        s = L.VattnStats()
        self._lib.vattn_get_stats(self._h, C.byref(s))
        return {n: int(getattr(s, n)) for n, _ in s._fields_}
