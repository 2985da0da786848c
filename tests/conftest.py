import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """HipKernels on cuda:0 — the product binding (no fallback: raises if the .so or the GPU is missing)."""
    import torch
    import sgg_amd  # noqa: F401
    from sgg_amd.lib import HipKernels
    assert torch.cuda.is_available(), "gpu-marked test started without a HIP device"
    return HipKernels("cuda:0")


@pytest.fixture(scope="session")
def ref():
    from oracle.kernels_ref import RefKernels
    return RefKernels()
