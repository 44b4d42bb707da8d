"""Install the MI355X-native modules under the names the reference's Python code imports:

    import vattention                      -> vattention_amd.vattention
    from flash_attn import flash_attn_with_kvcache, flash_attn_func -> vattention_amd.flash_attn
    from sarathi.cache_ops import cache_flat -> vattention_amd.cache_ops

so that sarathi-lean's own vattention_flashattention_wrapper.py / vATTN_cache_engine.py run unmodified
(see INTEGRATION.md).  Existing modules of those names are never overwritten unless force=True.
"""
from __future__ import annotations

import sys
import types


def install(force: bool = False) -> None:
    from . import cache_ops, flash_attn, vattention
    for name, mod in (("vattention", vattention), ("flash_attn", flash_attn)):
        if force or name not in sys.modules:
            sys.modules[name] = mod
    if force or "sarathi.cache_ops" not in sys.modules:
        sys.modules["sarathi.cache_ops"] = cache_ops
        pkg = sys.modules.get("sarathi")
        if pkg is None:
            pkg = types.ModuleType("sarathi")
            pkg.__path__ = []          # namespace-like stub; a real sarathi on sys.path takes precedence
            sys.modules["sarathi"] = pkg
        setattr(pkg, "cache_ops", cache_ops)
