"""Loads the REFERENCE's own attention wrapper and cache engine (sarathi-lean) against the MI355X drop-ins.  TEST INFRASTRUCTURE.

Source: the files in place under /root/reference when it exists (this container), else their byte-compiled form under
oracle/_ref/pyref (oracle/build_pyref.py; what travels to the GPU box).  The rest of sarathi's import closure (config, logger,
metrics, ray ...) is replaced by inert stubs; `vattention`, `flash_attn` and `sarathi.cache_ops` resolve to the drop-ins
(vattention_amd/dropin.py) — so every line of the reference's forward() / step() executes unmodified on the native stack.
"""
import importlib.machinery
import importlib.util
import logging
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
SRC = {
    "base_attention_wrapper": "sarathi-lean/sarathi/model_executor/attention/base_attention_wrapper.py",
    "vattention_flashattention_wrapper": "sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py",
    "base_cache_engine": "sarathi-lean/sarathi/worker/cache_engine/base_cache_engine.py",
    "vATTN_cache_engine": "sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py",
}
MODNAME = {
    "base_attention_wrapper": "sarathi.model_executor.attention.base_attention_wrapper",
    "vattention_flashattention_wrapper": "sarathi.model_executor.attention.vattention_flashattention_wrapper",
    "base_cache_engine": "sarathi.worker.cache_engine.base_cache_engine",
    "vATTN_cache_engine": "sarathi.worker.cache_engine.vATTN_cache_engine",
}


def available() -> str:
    """'source', 'pyc' or '' (neither present)."""
    if all(os.path.exists(os.path.join(REF, p)) for p in SRC.values()):
        return "source"
    if all(os.path.exists(os.path.join(ROOT, "oracle", "_ref", "pyref", n + ".pyc")) for n in SRC):
        return "pyc"
    return ""


def _load(name: str, how: str):
    modname = MODNAME[name]
    if how == "source":
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, SRC[name]))
    else:
        path = os.path.join(ROOT, "oracle", "_ref", "pyref", name + ".pyc")
        spec = importlib.util.spec_from_loader(modname, importlib.machinery.SourcelessFileLoader(modname, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


class loaded:
    """Context manager: installs stubs + drop-ins, loads the four reference modules, restores sys.modules on exit.
    `cpu_kernels`: (flash_attn_with_kvcache, cache_flat) callables to bind instead of the GPU drop-ins (golden generation on CPU)."""

    def __init__(self, cpu_kernels=None):
        self.cpu_kernels = cpu_kernels

    def __enter__(self):
        how = available()
        if not how:
            raise RuntimeError("reference wrapper/engine not available (neither /root/reference nor oracle/_ref/pyref)")
        if self.cpu_kernels is None:
            # (imported BEFORE sys.modules is saved: a module first imported inside the context would be dropped from sys.modules on exit
            # while the package still holds it as an attribute — the next `import vattention_amd.vattention` then builds a second copy)
            import vattention_amd.cache_ops, vattention_amd.dropin, vattention_amd.flash_attn, vattention_amd.vattention  # noqa: F401, E401
        self.saved = dict(sys.modules)
        for name in list(sys.modules):
            if name == "sarathi" or name.startswith("sarathi.") or name in ("vattention", "flash_attn"):
                sys.modules.pop(name)

        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            if "." not in name or True:
                m.__path__ = []
            sys.modules[name] = m
            return m

        if self.cpu_kernels is None:
            import vattention_amd.dropin as dropin
            dropin.install(force=True)
            import vattention_amd.vattention as va
            self.vattention = va
        else:
            fa, cf = self.cpu_kernels
            stub("flash_attn", flash_attn_with_kvcache=fa, flash_attn_func=None)
            stub("sarathi")
            stub("sarathi.cache_ops", cache_flat=cf)
            self.vattention = stub("vattention")
        from vattention_amd.attention.timers import OperationMetrics, OpTimer
        stub("sarathi.config", ModelConfig=object, ParallelConfig=object, CacheConfig=object)
        stub("sarathi.core"); stub("sarathi.core.datatypes")
        stub("sarathi.core.datatypes.sequence", SequenceMetadata=object, Sequence=object)
        stub("sarathi.logger", init_logger=lambda n: logging.getLogger(n))
        stub("sarathi.utils", in_wsl=lambda: False)
        stub("sarathi.metrics")
        stub("sarathi.metrics.constants", OperationMetrics=OperationMetrics)
        stub("sarathi.metrics.cuda_timer", CudaTimer=OpTimer)
        stub("sarathi.model_executor")
        holder = {}
        stub("sarathi.model_executor.attention", get_attention_wrapper=lambda: holder["wrapper"])
        stub("sarathi.worker"); stub("sarathi.worker.cache_engine")
        self.base_wrapper = _load("base_attention_wrapper", how)
        self.wrapper_mod = _load("vattention_flashattention_wrapper", how)
        self.base_engine = _load("base_cache_engine", how)
        self.engine_mod = _load("vATTN_cache_engine", how)
        self.wrapper_mod.VAttentionFlashAttentionWrapper._inst = None
        holder["wrapper"] = self.wrapper_mod.VAttentionFlashAttentionWrapper.get_instance()
        self.wrapper = holder["wrapper"]
        self.how = how
        return self

    def __exit__(self, *exc):
        sys.modules.clear()
        sys.modules.update(self.saved)
        return False
