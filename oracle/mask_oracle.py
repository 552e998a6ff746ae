"""CPU oracle for SURVEY.md section 8(f) N3: the post-PIV masks of ``ds.velocimetry.mask`` (pyorc/api/mask.py:147-403).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product path (pyorc_amd/mask.py calls the HIP library).

The reference writes these masks as xarray expressions over the float32 variables ``v_x, v_y, corr, s2n`` on
``(time, y, x)``.  xarray is not importable here, so each expression is restated with the numpy calls xarray dispatches
to when bottleneck / numbagg are absent (they are not pyorc dependencies): ``mean``/``std``/``count`` with
``skipna`` -> ``np.nanmean`` / ``np.nanstd`` (ddof 0) / ``count_nonzero(~isnan)``, ``shift`` -> NaN-filled shift,
``rolling(center=True).max()`` -> window ``[i - w//2, i + (w-1)//2]`` with NaN where it is incomplete, python-float
thresholds compared in float32 (NEP 50 weak scalars; same result under numpy 1.x value-based casting).
PINNED for corr, minmax, rolling, outliers, variance, count, window_mean (+ reduce_time, apply): the chain of the reference's
masking notebook applied to examples/ngwerere/ngwerere_piv.nc reproduces examples/ngwerere/ngwerere_masked.nc exactly
(tests/test_masks.py::test_mask_oracle_reproduces_the_reference_masked_file, fixture tests/golden/ngwerere_masks.npz).
angle, s2n, window_nan, window_replace: PARITY UNPINNED against a real xarray run (tests/test_mask.py holds no numbers).

Quirks reproduced because they change results:
  * ``helpers.stack_window`` (pyorc/helpers.py:672-679) iterates ``range(wdw_y_min, wdw_y_max)`` -- the +wdw_y_max row
    is NOT part of the neighbourhood (6 neighbours for wdw=1, not 9).
  * ``variance`` clamps the mean with ``np.maximum(mean, 1e30)`` (mask.py:268-269), so the mask is "std is not NaN".
  * ``rolling`` leaves NaN at the first w//2 and last (w-1)//2 time steps, which therefore mask out everything.
  * ``window_mean`` divides by the signed neighbourhood mean (negative mean -> always inside tolerance).

Fields are passed as a ``(4, T, R, C)`` float32 block ``[v_x, v_y, corr, s2n]`` (the layout of the device result).
"""

from __future__ import annotations

import warnings

import numpy as np

VX, VY, CORR, S2N = 0, 1, 2, 3
f32 = np.float32


def _quiet(fn, *a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        with np.errstate(all="ignore"):
            return fn(*a, **k)


def speed(f):
    """(v_x ** 2 + v_y ** 2) ** 0.5 -- numpy's scalar-power fast paths make this square, square, add, sqrt."""
    return np.sqrt(np.square(f[VX]) + np.square(f[VY]))


def minmax(f, s_min=0.1, s_max=5.0):                               # mask.py:147-160
    s = speed(f)
    return (s > f32(s_min)) & (s < f32(s_max))


def angle(f, angle_expected=0.5 * np.pi, angle_tolerance=0.25 * np.pi):   # mask.py:162-185
    a = np.arctan2(f[VX], f[VY])
    return _quiet(lambda: np.abs(a - f32(angle_expected)) < f32(angle_tolerance))


def count(f, tolerance=0.33):                                      # mask.py:187-201 -> (R, C)
    return np.count_nonzero(~np.isnan(f[VX]), axis=0) > tolerance * f.shape[1]


def corr(f, tolerance=0.1):                                        # mask.py:203-213
    return _quiet(lambda: f[CORR] > f32(tolerance))


def s2n(f, tolerance=10):                                          # mask.py:215-225
    return _quiet(lambda: f[S2N] > f32(tolerance))


def outliers(f, tolerance=1.0, mode="or"):                         # mask.py:227-252
    def cond(v):
        std, mean = _quiet(np.nanstd, v, axis=0), _quiet(np.nanmean, v, axis=0)
        return _quiet(lambda: np.abs((v - mean) / std) < f32(tolerance))
    x, y = cond(f[VX]), cond(f[VY])
    return x | y if mode == "or" else x & y


def variance(f, tolerance=5, mode="and"):                          # mask.py:254-284 -> (R, C)
    def cond(v):
        std = _quiet(np.nanstd, v, axis=0)
        mean = _quiet(lambda: np.maximum(_quiet(np.nanmean, v, axis=0), f32(1e30)))
        return _quiet(lambda: np.abs(std / mean) < f32(tolerance))
    x, y = cond(f[VX]), cond(f[VY])
    return x | y if mode == "or" else x & y


def rolling(f, wdw=5, tolerance=0.5):                              # mask.py:286-303
    s = _quiet(speed, f)
    s0 = np.where(np.isnan(s), f32(0), s)
    T = s.shape[0]
    roll = np.full_like(s, np.nan)
    lo, hi = wdw // 2, (wdw - 1) // 2
    for t in range(lo, T - hi):
        roll[t] = s0[t - lo:t + hi + 1].max(axis=0)
    return _quiet(lambda: s > f32(tolerance) * roll)


def _strides(wdw=1, wdw_x_min=None, wdw_x_max=None, wdw_y_min=None, wdw_y_max=None):
    x_min = -wdw if wdw_x_min is None else wdw_x_min
    x_max = wdw if wdw_x_max is None else wdw_x_max
    y_min = -wdw if wdw_y_min is None else wdw_y_min
    y_max = wdw if wdw_y_max is None else wdw_y_max
    return [(sx, sy) for sx in range(x_min, x_max + 1) for sy in range(y_min, y_max)]   # helpers.py:672-679


def _shift(a, sx, sy):
    """xarray ``shift(x=sx, y=sy)``: out[..., r, c] = a[..., r - sy, c - sx], NaN where that falls outside."""
    out = np.full_like(a, np.nan)
    R, C = a.shape[-2:]
    r0, r1 = max(sy, 0), min(R + sy, R)
    c0, c1 = max(sx, 0), min(C + sx, C)
    if r0 < r1 and c0 < c1:
        out[..., r0:r1, c0:c1] = a[..., r0 - sy:r1 - sy, c0 - sx:c1 - sx]
    return out


def stack_window(a, **kw):
    return np.stack([_shift(a, sx, sy) for sx, sy in _strides(**kw)])


def window_nan(f, tolerance=0.7, wdw=1, **kw):                     # mask.py:305-340 (applied per time step)
    st = stack_window(f[VX], wdw=wdw, **kw)
    if st.shape[0] == 0:
        return np.zeros(f[VX].shape, bool) >= 0
    return np.count_nonzero(~np.isnan(st), axis=0) >= tolerance * st.shape[0]


def window_mean(f, tolerance=0.7, wdw=1, mode="or", **kw):         # mask.py:342-383
    def cond(v):
        m = _quiet(np.nanmean, stack_window(v, wdw=wdw, **kw), axis=0)
        return _quiet(lambda: np.abs(v - m) / m < f32(tolerance))
    x, y = cond(f[VX]), cond(f[VY])
    return x | y if mode == "or" else x & y


def window_replace(f, wdw=1, iter=1, **kw):                        # mask.py:385-403 -> new (4, T, R, C) block
    f = f.copy()
    for _ in range(iter):
        for k in range(4):
            m = _quiet(np.nanmean, stack_window(f[k], wdw=wdw, **kw), axis=0)
            f[k] = np.where(np.isnan(f[k]), m, f[k])
    return f


def time_mean(f):
    """``ds.mean(dim="time")`` (the ``reduce_time=True`` pre-step, mask.py:51-52) -> (4, 1, R, C)."""
    return _quiet(np.nanmean, f, axis=1)[:, None]


def apply(f, mask):
    """``ds[var].where(mask)`` on all four variables (mask.py:132-145); mask (T,R,C) or (R,C)."""
    return np.where(np.broadcast_to(mask, f.shape), f, f32(np.nan)).astype(f32)


def scale_velocity(u, v, res_x, res_y, dt):
    """ffpiv.py:418-419: ``(u * res_x / dt[:, None, None]).astype(float32)`` with python-float resolutions."""
    dt = np.asarray(dt, dtype=np.float64)[:, None, None]
    return (u * float(res_x) / dt).astype(f32), (v * float(res_y) / dt).astype(f32)
