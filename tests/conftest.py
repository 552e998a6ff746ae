import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def lib():
    from pyorc_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def gpu(lib):
    from pyorc_amd import _lib

    _lib.require_device()  # loud failure instead of a silent fallback
    return lib
