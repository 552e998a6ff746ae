"""Mirror of ``ds.velocimetry.mask`` (pyorc/api/mask.py) on the MI355X: SURVEY.md section 8(f) N3.

``Mask(ds)`` wraps a PIV result -- a ``velocimetry.PivResult`` / dict / ``xarray.Dataset`` holding ``v_x, v_y, corr,
s2n`` as float32 ``(time, y, x)`` or ``(y, x)`` arrays -- and offers the reference's mask methods with the reference's
names, defaults, ``inplace`` / ``reduce_time`` switches, assertions and warnings (the ``_base_mask`` wrapper,
mask.py:22-90).  Each mask is one kernel over the ``[v_x | v_y | corr | s2n]`` block (``lspiv_mask``, csrc/masks.hip);
there is no CPU fallback.  Masks come back as boolean numpy arrays (``xarray.DataArray`` when the input was a Dataset).

For results that are still in HBM use the ``*_dev`` entry points directly (``lspiv_scale_velocity_dev`` ->
``lspiv_mask_dev`` ... -> ``lspiv_mask_apply_dev`` -> ``lspiv_pack_int16_dev``); tools/mask_bench.py shows the chain.
"""

from __future__ import annotations

import warnings
from typing import Optional

import numpy as np

from . import _lib

VARS = ("v_x", "v_y", "corr", "s2n")
KINDS = {"minmax": 0, "angle": 1, "count": 2, "corr": 3, "s2n": 4, "outliers": 5, "variance": 6, "rolling": 7,
         "window_nan": 8, "window_mean": 9}
TIME_MSG = ('This mask requires dimension "time". The dataset does not contain dimension "time" or you '
            "have set `reduce_time=True`. Apply this mask without applying any reducers in time.")
MULTI_MSG = ("This mask requires multiple timesteps in the dataset in order have an effect. This "
             "warning typically occurs when applying `Frames.get_piv(ensemble_corr=True)` as this only "
             "yields one single time step.")

try:
    import xarray as xr
except ImportError:  # pragma: no cover - depends on the environment
    xr = None


def _mode(mode: str) -> float:
    return 0.0 if mode == "or" else 1.0


def _window(wdw=1, wdw_x_min=None, wdw_x_max=None, wdw_y_min=None, wdw_y_max=None):
    """helpers.stack_window's stride resolution (pyorc/helpers.py:667-670)."""
    return [float(-wdw if wdw_x_min is None else wdw_x_min), float(wdw if wdw_x_max is None else wdw_x_max),
            float(-wdw if wdw_y_min is None else wdw_y_min), float(wdw if wdw_y_max is None else wdw_y_max)]


def fields_block(ds) -> tuple[np.ndarray, bool]:
    """(4, T, R, C) float32 block + whether the variables carried a time axis."""
    arrs = [np.asarray(ds[k].values if hasattr(ds[k], "values") else ds[k], dtype=np.float32) for k in VARS]
    has_time = arrs[0].ndim == 3
    if arrs[0].ndim not in (2, 3) or any(a.shape != arrs[0].shape for a in arrs):
        raise AssertionError("Dataset is not a valid velocimetry dataset")
    block = np.stack([a if has_time else a[None] for a in arrs])
    return np.ascontiguousarray(block), has_time


def run_mask(block: np.ndarray, kind: str, params) -> np.ndarray:
    """One ``lspiv_mask`` call on a host block -> bool (T, R, C) or (R, C)."""
    _lib.require_device()
    _, T, R, C = block.shape
    k = KINDS[kind]
    p = np.asarray(params, dtype=np.float64)
    out = np.empty((R, C) if kind in ("count", "variance") else (T, R, C), dtype=np.uint8)
    _lib.check(_lib.load().lspiv_mask(_lib.ptr(block), T, R, C, k, _lib.ptr(p), len(p), _lib.ptr(out)))
    return out.astype(bool)


def time_mean(block: np.ndarray) -> np.ndarray:
    """``ds.mean(dim="time")`` of the four variables -> (4, 1, R, C)."""
    _lib.require_device()
    _, T, R, C = block.shape
    out = np.empty((4, 1, R, C), dtype=np.float32)
    _lib.check(_lib.load().lspiv_time_mean(_lib.ptr(block), T, R, C, _lib.ptr(out)))
    return out


def apply_mask(block: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """``where(mask)`` on all four variables of a host block (returns a new block)."""
    _lib.require_device()
    _, T, R, C = block.shape
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if m.shape not in ((T, R, C), (R, C)):
        raise ValueError(f"mask shape {m.shape} does not fit fields {(T, R, C)}")
    out = block.copy()
    _lib.check(_lib.load().lspiv_mask_apply(_lib.ptr(out), T, R, C, _lib.ptr(m), int(m.ndim == 3)))
    return out


class Mask:
    """``ds.velocimetry.mask`` for ``engine="hip"`` results."""

    def __init__(self, ds):
        self._obj = ds

    # -- the _base_mask wrapper (mask.py:22-90) ----------------------------------------------------------------
    def _run(self, kind, params, inplace, reduce_time, time_allowed=False, time_required=False, multi=False):
        block, has_time = fields_block(self._obj)
        if reduce_time and has_time:
            block, has_time = time_mean(block), False
        n_t = block.shape[1]
        if time_required:
            if not has_time:
                raise AssertionError(TIME_MSG)
            if multi and n_t < 2:
                warnings.warn(MULTI_MSG, stacklevel=3)
        if multi and time_required and n_t < 2:
            mask = np.ones(block.shape[2:], dtype=bool)            # just pass Trues everywhere
        else:
            mask = run_mask(block, kind, params)
            if not has_time and mask.ndim == 3:
                mask = mask[0]
        if inplace:
            self._store(apply_mask(fields_block(self._obj)[0], mask))
        return self._wrap(mask)

    def _wrap(self, mask):
        if xr is not None and isinstance(self._obj, xr.Dataset):
            dims = ("time", "y", "x") if mask.ndim == 3 else ("y", "x")
            return xr.DataArray(mask, dims=dims, coords={d: self._obj[d] for d in dims})
        return mask

    def _store(self, block):
        has_time = np.asarray(self._obj["v_x"]).ndim == 3
        for i, k in enumerate(VARS):
            new = block[i] if has_time else block[i, 0]
            if xr is not None and isinstance(self._obj, xr.Dataset):
                # copy(data=...) keeps the variable's attrs and encoding (units, standard_name, the int16 scale / fill
                # of set_encoding), which a (dims, ndarray) assignment would drop
                self._obj[k] = self._obj[k].copy(data=new.reshape(self._obj[k].shape))
            else:
                self._obj[k] = new

    def __call__(self, mask, inplace=False):
        """Apply one mask or a list of masks (mask.py:111-145); returns the masked copy unless ``inplace``."""
        masks = mask if isinstance(mask, list) else [mask]
        block, _ = fields_block(self._obj)
        for m in masks:
            block = apply_mask(block, np.asarray(m.values if hasattr(m, "values") else m))
        if inplace:
            self._store(block)
            return None
        import copy

        out = Mask(copy.deepcopy(self._obj))
        out._store(block)
        return out._obj

    # -- the masks, names / defaults as in the reference --------------------------------------------------------
    def minmax(self, inplace=False, reduce_time=False, s_min=0.1, s_max=5.0):
        return self._run("minmax", [s_min, s_max], inplace, reduce_time, time_allowed=True)

    def angle(self, inplace=False, reduce_time=False, angle_expected=0.5 * np.pi, angle_tolerance=0.25 * np.pi):
        return self._run("angle", [angle_expected, angle_tolerance], inplace, reduce_time, time_allowed=True)

    def count(self, inplace=False, reduce_time=False, tolerance=0.33):
        return self._run("count", [tolerance], inplace, reduce_time, time_required=True, multi=True)

    def corr(self, inplace=False, reduce_time=False, tolerance=0.1):
        return self._run("corr", [tolerance], inplace, reduce_time, time_allowed=True)

    def s2n(self, inplace=False, reduce_time=False, tolerance=10):
        return self._run("s2n", [tolerance], inplace, reduce_time, time_allowed=True)

    def outliers(self, inplace=False, reduce_time=False, tolerance=1.0, mode="or"):
        return self._run("outliers", [tolerance, _mode(mode)], inplace, reduce_time, time_required=True, multi=True)

    def variance(self, inplace=False, reduce_time=False, tolerance=5, mode="and"):
        return self._run("variance", [tolerance, _mode(mode)], inplace, reduce_time, time_required=True, multi=True)

    def rolling(self, inplace=False, reduce_time=False, wdw=5, tolerance=0.5):
        return self._run("rolling", [wdw, tolerance], inplace, reduce_time, time_required=True, multi=True)

    def window_nan(self, inplace=False, reduce_time=False, tolerance=0.7, wdw=1, **kwargs):
        return self._run("window_nan", [tolerance] + _window(wdw, **kwargs), inplace, reduce_time)

    def window_mean(self, inplace=False, reduce_time=False, tolerance=0.7, wdw=1, mode="or", **kwargs):
        return self._run("window_mean", [tolerance, _mode(mode)] + _window(wdw, **kwargs), inplace, reduce_time)

    def window_replace(self, inplace=False, reduce_time=False, wdw=1, iter=1, **kwargs):
        """Returns a result (not a mask) with NaNs replaced by their neighbourhood mean (mask.py:385-403)."""
        _lib.require_device()
        block, has_time = fields_block(self._obj)
        if reduce_time and has_time:
            block, has_time = time_mean(block), False
        block = block.copy()
        w = [int(v) for v in _window(wdw, **kwargs)]
        _lib.check(_lib.load().lspiv_window_replace(_lib.ptr(block), block.shape[1], block.shape[2], block.shape[3], *w, int(iter)))
        import copy

        out = Mask(copy.deepcopy(self._obj))
        if has_time or np.asarray(self._obj["v_x"]).ndim == 2:
            out._store(block)
        else:  # time was reduced away: the result has no time axis any more
            for i, k in enumerate(VARS):
                out._obj[k] = block[i, 0]
        return out._obj
