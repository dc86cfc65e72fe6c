import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # (pytest-timeout's marker, registered here too so that the suite also collects cleanly without the plugin)
    config.addinivalue_line("markers", "timeout(seconds): fail the test after this many seconds (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when collected on a box without a GPU and without -m gpu."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
