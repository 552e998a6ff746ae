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
    """32 x 32 windows: select the per-pair kernel (LSPIV_WALK=0), whose results do not depend on how the time axis is
    chunked -- bit for bit.  The default time-walking kernel shares transforms between consecutive frames of a window,
    so two different chunkings of the same stack agree to float32 rounding only (see `assert_chunk_close`)."""
    monkeypatch.setenv("LSPIV_WALK", "0")


def assert_chunk_close(part, whole, corr_tol=1e-5, uv_tol=1e-4):
    """Results of the same frame pairs from two different chunkings of a stack, default (time-walking) kernel.
    part / whole: sequences (u, v, corr, s2n) of equal shape.  Same NaN mask up to arg-max ties on a plane border,
    corr / s2n to 1e-5; displacements, wherever both runs picked the same peak: 99.9 % within 1e-4 of
    max(|ref|, 0.05 px) -- the rest are the ill-conditioned sub-pixel fits (flat ridges, empty neighbours) that
    amplify float32 rounding for any implementation (oracle.c_oracle.well_posed grades them in the parity tests)."""
    import numpy as np

    u, v, c, s = (np.asarray(a, dtype=np.float64) for a in part)
    uo, vo, co, so = (np.asarray(a, dtype=np.float64) for a in whole)
    assert u.shape == uo.shape
    assert np.array_equal(np.isnan(c), np.isnan(co)) and np.array_equal(np.isnan(s), np.isnan(so))
    assert (np.isnan(u) != np.isnan(uo)).mean() < 1e-4
    with np.errstate(all="ignore"):
        assert np.nanmax(np.abs(c - co) / np.maximum(np.abs(co), 0.05), initial=0.0) <= corr_tol
        assert np.nanmax(np.abs(s - so) / np.maximum(np.abs(so), 0.05), initial=0.0) <= corr_tol
        same = (np.abs(u - uo) < 0.5) & (np.abs(v - vo) < 0.5)
        assert same[~np.isnan(u) & ~np.isnan(uo)].mean() > 0.999
        for g, r in ((u, uo), (v, vo)):
            e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[same]
            e = e[~np.isnan(e)]
            assert e.size == 0 or (np.percentile(e, 99.9) <= uv_tol and e.max() <= 0.2)
