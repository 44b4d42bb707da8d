"""vattention_amd — MI355X-native vAttention hot path.

Sub-modules:
  page_manager   object wrapper over the C-ABI page manager (include/vattn.h)
  vattention     drop-in for the reference's `vattention` Python module
  flash_attn     drop-in for the `flash_attn` calls the reference's wrappers make
  cache_ops      drop-in for `sarathi.cache_ops.cache_flat`
  attention      mirror of sarathi-lean's attention-wrapper API (fa_vattn backend)
  cache_engine   mirror of sarathi-lean's vATTNCacheEngine
  dropin         installs the drop-ins under their reference names in sys.modules
"""
__all__ = ["page_manager"]
