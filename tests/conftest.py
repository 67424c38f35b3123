import os
import sys

import pytest

# PyTorch-ROCm bundles its own HIP runtime: in a process that uses both, torch has to be imported BEFORE libapss_hip.so
# pulls in the system's libamdhip64 (two runtimes in one process: the second one finds no device).  Some GPU tests use
# torch for device tensors, so import it up front whatever subset of the tests is selected.
try:
    import torch  # noqa: F401
except ImportError:  # the CPU-only oracle tests do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "all-pairs-similarity_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
