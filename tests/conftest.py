import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """torch brings its own copy of the HIP runtime; in one process it has to open the GPU BEFORE the library (which links the
    system's copy) does -- the other way round torch reports "No HIP GPUs are available" (seen when a test file whose first tests
    use the library alone is run on its own). Where there is no GPU this is a no-op."""
    try:
        import torch
        if torch.cuda.device_count() > 0 and torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
