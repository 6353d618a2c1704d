import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """Build (only if missing or stale) the in-tree HIP library before any test runs; hipcc
    cross-compiles gfx950 without a GPU, so this works in the CPU container and on the GPU box."""
    from ml2048_amd import build

    build.build()
