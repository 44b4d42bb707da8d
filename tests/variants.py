"""Which `variant` values a GPU test may use.  `pytest -m "gpu and not lab"` is the PRODUCT-ONLY run (conftest.py then sets VATTN_NO_LAB): tests
that loop over variants drop the ones only tools/lab/libvattn_lab.so contains, and tools/lab/libvattn_lab.so is never loaded.  `pytest -m gpu`
(the driver's run) keeps everything; parametrized lab variants carry the `lab` marker (added at collection, conftest.py)."""
import os


def no_lab() -> bool:
    return os.environ.get("VATTN_NO_LAB", "") == "1"


def needs_lab(v: int) -> bool:
    from vattention_amd import kernels as K
    return K.needs_lab(int(v))


def enabled(*vs):
    """the variants of `vs` this run may launch, in order"""
    return [v for v in vs if not (no_lab() and needs_lab(v))]


def product_or_self(v: int) -> int:
    """a fuzz test's random variant: itself, or — in the product-only run — the product variant of the same tiling family"""
    if not (no_lab() and needs_lab(v)):
        return v
    keep = v & ((7 << 1) | (3 << 5) | (1 << 7))
    til = (keep >> 1) & 7
    if til not in (0, 1, 4, 7):
        keep = (keep & ~(7 << 1)) | ((7 if til in (2, 6) else 0) << 1)
    return keep
