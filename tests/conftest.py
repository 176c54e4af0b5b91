import os
import sys

import pytest

# PyTorch-ROCm wheels bundle their own libamdhip64; librtrace_hip.so links the system one.  Both have the same SONAME,
# so whichever is loaded first serves the whole process -- and torch only works on its own.  Tests that hand torch
# tensors / streams to the C ABI therefore need torch imported BEFORE the backend library is first loaded.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
