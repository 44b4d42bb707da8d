"""Where /root/reference is present (build container): the reference's OWN fa_vattn wrapper and cache-engine source
files import against the drop-ins (vattention / flash_attn / sarathi.cache_ops) with inert stubs for the rest of
sarathi's import closure, and bind to the MI355X-native symbols.  Nothing is copied; the files are loaded in place."""
import importlib.util
import os
import sys
import types

import pytest

REF = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
WRAP = os.path.join(REF, "sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py")


@pytest.mark.skipif(not os.path.exists(WRAP), reason="reference tree not present")
def test_reference_wrapper_imports_against_dropins():
    saved = dict(sys.modules)
    try:
        import vattention_amd.dropin as dropin
        for name in ("vattention", "flash_attn", "sarathi", "sarathi.cache_ops"):
            sys.modules.pop(name, None)
        dropin.install(force=True)

        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m
        stub("sarathi.config", ModelConfig=object, ParallelConfig=object, CacheConfig=object)
        stub("sarathi.core"); stub("sarathi.core.datatypes")
        stub("sarathi.core.datatypes.sequence", SequenceMetadata=object, Sequence=object)
        stub("sarathi.logger", init_logger=lambda n: __import__("logging").getLogger(n))
        stub("sarathi.metrics")
        from vattention_amd.attention.timers import OperationMetrics, OpTimer
        stub("sarathi.metrics.constants", OperationMetrics=OperationMetrics)
        stub("sarathi.metrics.cuda_timer", CudaTimer=OpTimer)
        stub("sarathi.model_executor"); stub("sarathi.model_executor.attention")
        base_path = os.path.join(os.path.dirname(WRAP), "base_attention_wrapper.py")
        spec = importlib.util.spec_from_file_location("sarathi.model_executor.attention.base_attention_wrapper", base_path)
        base = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = base
        spec.loader.exec_module(base)
        spec = importlib.util.spec_from_file_location("ref_vattn_fa_wrapper", WRAP)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        import vattention_amd.cache_ops as co
        import vattention_amd.flash_attn as fa
        import vattention_amd.vattention as va
        assert mod.flash_attn_with_kvcache is fa.flash_attn_with_kvcache
        assert mod.cache_flat is co.cache_flat
        assert mod.vattention is va
        w = mod.VAttentionFlashAttentionWrapper.get_instance()
        for meth in ("init", "begin_forward", "set_batch_idx", "forward", "end_forward"):
            assert callable(getattr(w, meth))
        # the 13 functions the reference module exports (vattention.cu:614-637) all exist on the drop-in
        for fn in ("reserve_physical_pages", "init_kvcache", "cleanup", "set_verbose", "set_deferred_reclamation",
                   "show_kvcache_config", "show_allocator_state", "step", "map_common_pages", "step_async",
                   "alloc_new_batch_idx", "free_batch_idx", "num_free_kvblocks"):
            assert callable(getattr(va, fn))
    finally:
        sys.modules.clear()
        sys.modules.update(saved)


def test_backend_registry_resolves_native_and_hybrid_backends():
    """attention/__init__.py:36-201 mirror: the fa_vattn family resolves to the native wrapper, fa_streams / fa_pod to the
    two-stream hybrid wrapper, CUDA-library backends fail by name (no silent substitution)."""
    import pytest
    from vattention_amd import attention as A
    prev = A.ATTENTION_BACKEND
    try:
        for name in ("fa_vattn", "FA_VATTN_SYNC", "fa_vattn_megacache"):
            A.set_attention_backend(name)
            assert type(A.get_attention_wrapper()).__name__ == "VAttentionFlashAttentionWrapper"
            assert A.is_vattention_backend() and not A.is_vLLM_backend()
        for name, cls in (("fa_streams", "VAttentionFlashAttentionStreamsWrapper"), ("fa_pod", "VAttentionFlashAttentionPodWrapper"),
                          ("fa_pod_megacache", "VAttentionFlashAttentionPodWrapper")):
            A.set_attention_backend(name)
            assert type(A.get_attention_wrapper()).__name__ == cls
            assert A.is_vattention_backend()
        for name in ("fi_vattn", "fa3_vattn", "fa_paged"):
            A.set_attention_backend(name)
            with pytest.raises(NotImplementedError):
                A.get_attention_wrapper()
        with pytest.raises(ValueError):
            A.set_attention_backend("no_such_backend")
    finally:
        A.ATTENTION_BACKEND = prev
