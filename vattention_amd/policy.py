"""KV-cache layout policy of the MI355X cache engine (no GPU needed).

One `hipMemCreate` handle backs one page, and creating a handle costs O(live handles) on ROCm 7.2 (9 us at 5 k handles, 106 us at
20 k, 0.9 ms at 100 k: profiles/r01_vmm_scale_probe.txt; a 259 GB pool of 2 MiB pages takes 98-125 s to fill).  The
reference creates every handle up front (vattention/cudaInternal.h:45-59) and offers small pages through a patched driver
(uvmInternal.h:146-217); here a pool that would need more than HANDLE_LIMIT handles is moved to the megacache layout — one page
covers all layers, 2 handles per page-group instead of 2*L (vattention.cu:42,54,146-147) — with pages of at least AUTO_PAGE bytes.
"""
from __future__ import annotations

from typing import Tuple

HANDLE_LIMIT = 100_000
AUTO_PAGE = 8 << 20


def choose_layout(page_size: int, megacache: bool, cache_mem_size: int, keep: bool = False) -> Tuple[int, bool, str]:
    """-> (page_size, megacache, description).  `keep` (cache_config.vattn_keep_layout / VATTN_KEEP_LAYOUT=1) keeps the configured
    layout whatever it costs."""
    pages = cache_mem_size // max(1, page_size)
    if keep or pages <= HANDLE_LIMIT:
        return page_size, megacache, "configured"
    new_page = max(page_size, AUTO_PAGE)
    return new_page, True, "auto: megacache, %d KiB pages, %d handles (configured: %s, %d KiB pages, %d handles)" % (
        new_page >> 10, cache_mem_size // new_page, "megacache" if megacache else "per-layer", page_size >> 10, pages)
