import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library bound to GPU 0. GPU tests fail loudly (no CPU fallback) if it is absent."""
    # torch (used by a few GPU tests for device tensors) ships its own HIP runtime: initialise it BEFORE
    # libicicle_hip.so pulls in /opt/rocm's, exactly as bench.py does, so the process has one runtime
    try:
        import torch

        torch.cuda.is_available() and torch.cuda.init()
    except Exception:
        pass
    import icicle_amd
    from icicle_amd import runtime

    assert runtime.get_device_count() >= 1, "no HIP device visible"
    runtime.set_device(0)
    return icicle_amd
