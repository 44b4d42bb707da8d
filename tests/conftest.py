import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock comparisons on a real MI355X (run with -m perf; never part of -m gpu: a slow box must not fail the correctness suite)")


def pytest_collection_modifyitems(config, items):
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
