import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: runs kernels of tools/lab/libvattn_lab.so (measurement builds, closed experiments) — NOT the product library; "
                                        "`-m \"gpu and not lab\"` is the product-only GPU suite")
    # the product-only run: tests that loop over variants drop the lab-only ones (tests/variants.py)
    if "not lab" in (config.getoption("-m") or ""):
        os.environ["VATTN_NO_LAB"] = "1"
    config.addinivalue_line("markers", "perf: wall-clock comparisons on a real MI355X (run with -m perf; never part of -m gpu: a slow box must not fail the correctness suite)")


def _mark_lab(items):
    """a parametrized `variant` (or `variant`-like id) that only the lab library contains makes the item a lab test"""
    try:
        from vattention_amd import kernels as K
    except Exception:
        return
    for item in items:
        cs = getattr(item, "callspec", None)
        if cs is None:
            continue
        for name in ("variant", "dvar"):
            v = cs.params.get(name)
            if isinstance(v, int) and K.needs_lab(v):
                item.add_marker(pytest.mark.lab)


def pytest_collection_modifyitems(config, items):
    _mark_lab(items)
    # GPU tests are selected explicitly with -m gpu; without a device they are skipped, not failed.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords or "perf" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """which in-tree native libraries this pytest process mapped: the product-only GPU run (-m "gpu and not lab") must not list tools/lab/"""
    try:
        libs = sorted({line.split()[-1] for line in open("/proc/self/maps") if line.rstrip().endswith(".so") and ROOT in line})
    except OSError:
        return
    if libs:
        terminalreporter.write_line("in-tree native libraries mapped: " + ", ".join(os.path.relpath(x, ROOT) for x in libs))
