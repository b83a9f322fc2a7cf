import os
import sys
import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # as plen_ml_walk_amd/__init__.py: before the HIP runtime initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it (gcc, seconds) before any test needs it."""
    from oracle import oracle
    oracle.build()
