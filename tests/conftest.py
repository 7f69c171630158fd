import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session", autouse=True)
def torch_hip_runtime_first():
    """torch ships its own HIP runtime; when the system runtime (libaardvark_amd.so links /opt/rocm's) has initialised the GPU first,
    torch.cuda reports "No HIP GPUs are available".  The GPU tests that hand torch tensors to the library therefore need torch's runtime
    to come up before any test touches the library.  device_count() does not initialise anything, so CPU-only runs are unaffected."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass
    yield
