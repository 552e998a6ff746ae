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


@pytest.fixture
def per_pair_kernel(monkeypatch):
    """Select the per-pair kernels (LSPIV_WALK=0): every window pair is computed on its own, so results do not depend on
    ANY chunking (the default time-walking kernels give chunk-independent results for chunks cut on multiples of
    lspiv_chunk_alignment pairs -- which is what get_ffpiv, the host entry points and pyorc_amd.shard do)."""
    monkeypatch.setenv("LSPIV_WALK", "0")
