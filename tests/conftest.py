import os
import sys

import pytest

# PyTorch bundles its own HIP runtime: it has to be loaded before libpm_gpu.so pulls in the system one,
# or torch finds no GPU later in the same process (tests that allocate the stream with torch).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.fixture(scope="session")
def ref_harness():
    if not os.path.exists(REF_HARNESS):
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference)")
    return REF_HARNESS
