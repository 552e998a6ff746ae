"""ctypes wrapper of oracle/libpiv_oracle.so (the plain-C, OpenMP restatement).  TEST INFRASTRUCTURE ONLY.

** PARITY UNPINNED ** -- see oracle/piv_oracle.py.  Used by tests (validated against the numpy oracle)
and by bench.py's cpu_baseline leg; never by pyorc_amd.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpiv_oracle.so")
_lib = None
_CODES = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        lib = C.CDLL(_SO)
        lib.piv_oracle_pairs.restype = C.c_int
        lib.piv_oracle_pairs.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int]
        lib.piv_oracle_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def max_threads() -> int:
    return load().piv_oracle_max_threads()


def piv_pairs(frames, window_size, overlap, signal_threshold=None, return_planes=False, nthreads=0,
              return_cond=False):
    """(u, v, corr_max, s2n[, planes][, cond]) for every consecutive pair of frames (T,H,W); u, v in pixels.

    cond (T-1, n_rows, n_cols, 3): [..., 0] relative gap between the plane maximum and the runner-up,
    [..., 1] smallest peak-neighbourhood value / maximum, [..., 2] smallest log-curvature of the peak --
    see ``well_posed``."""
    a = np.ascontiguousarray(frames)
    if a.dtype not in _CODES:
        a = a.astype(np.float64)
    T, H, W = a.shape
    wy, wx = window_size
    oy, ox = overlap
    n_rows = (H - wy) // (wy - oy) + 1
    n_cols = (W - wx) // (wx - ox) + 1
    out = [np.empty((T - 1, n_rows, n_cols), dtype=np.float32) for _ in range(4)]
    planes = np.empty((T - 1, n_rows * n_cols, wy, wx), dtype=np.float64) if return_planes else None
    thr = -1.0 if signal_threshold is None else float(signal_threshold)
    cond = np.zeros((T - 1, n_rows, n_cols, 3), dtype=np.float32) if return_cond else None
    rc = load().piv_oracle_pairs(a.ctypes.data, _CODES[a.dtype], T, H, W, wy, wx, oy, ox, thr,
                                 out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data,
                                 planes.ctypes.data if planes is not None else None,
                                 cond.ctypes.data if cond is not None else None, int(nthreads))
    if rc != 0:
        raise ValueError("piv_oracle_pairs: bad arguments")
    res = list(out)
    if return_planes:
        res.append(planes)
    if return_cond:
        res.append(cond)
    return tuple(res)


def unique_peak(cond, min_gap=1e-5):
    """Windows whose arg-max (hence whether the peak is on the border => NaN) is unique under float32 noise."""
    return cond[..., 0] >= min_gap


def well_posed(cond, min_gap=1e-5, min_neighbour=0.02, min_curvature=0.05):
    """Windows whose sub-pixel result is stable under float32 rounding of the correlation plane.

    A plane is computed to ~1e-7 of its maximum in float32.  The arg-max is unique under that noise when the
    runner-up is >= ``min_gap`` below the maximum, and the log-Gaussian fit moves by < 1e-5 px when every
    neighbour of the peak is >= ``min_neighbour`` of the maximum and the peak is not a flat ridge (log-curvature
    2 ln c- - 4 ln c0 + 2 ln c+ at least ``min_curvature`` in both directions: the fit divides by it).  Ill-posed
    windows (empty / single-speckle windows at the frame edge, flat ridges) are excluded from the 1e-4 parity
    gate on u, v and counted separately; their NaN mask, corr_max and s2n are still gated.
    """
    return (cond[..., 0] >= min_gap) & (cond[..., 1] >= min_neighbour) & (cond[..., 2] >= min_curvature)


def exact_tie(cond, corr_max, tol=1e-12):
    """Windows whose float64 plane maximum is not unique: the runner-up is within ``tol`` (relative) of it, or the whole
    plane is below ``tol`` (zero in exact arithmetic -- the normalised windows are >= 0 -- so that every sample ties).

    Which of two equal samples is "the" arg-max is decided by the rounding of whatever transform computed them; the
    oracle's own answer there is not reproducible by any other implementation (ffpiv's FFT included).  Since round 3
    these are the ONLY windows the 1e-4 gate on u, v sets aside: the library's float64 rescue pass covers the
    ill-conditioned ones that ``well_posed`` used to exclude."""
    cm = np.asarray(corr_max)
    with np.errstate(invalid="ignore"):
        return ((cond[..., 0] < tol) | (cm < tol)) & (cm > 0)
