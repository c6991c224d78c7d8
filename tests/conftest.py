import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """On a box without an MI355X the gpu-marked tests are skipped (plain `pytest tests` stays green); on a GPU box
    they run and fail loudly if the HIP library is missing -- there is no fallback to hide behind."""
    try:
        import torch
        have_gpu = torch.cuda.is_available() and torch.cuda.device_count() > 0
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_ctx():
    """Shared device context for the -m gpu tier; fails loudly when the HIP library or GPU is missing."""
    from moleculekit_amd import _lib

    return _lib.default_context()
