"""Mirror of the element-wise ``Frames`` filters of pyorc on the MI355X.

``normalize`` (pyorc/api/frames.py:279-306), ``minmax`` (:344-362) and ``time_diff`` (:409-436) are plain
numpy/xarray arithmetic in the reference and are reproduced bit for bit.  ``smooth`` (:438-467) and ``edge_detect``
(:308-342) are ``cv2.GaussianBlur`` calls (pyorc/cv.py:142-183): OpenCV cannot be installed here, so its published
algorithm is restated (coefficient tables, separable float32 filter, BORDER_REFLECT_101) -- expect ~1e-6 agreement
with a real cv2, not bit identity.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from .device import DeviceFrames, is_device


def time_diff(frames, thres: float = 0.0, abs: bool = False) -> np.ndarray:
    """``Frames.time_diff``: (T, H, W) -> (T-1, H, W) float32; values <= thres (and NaN) become 0.
    A ``DeviceFrames`` stack stays in HBM (``lspiv_time_diff_dev``) -- so do all the filters below."""
    if is_device(frames):
        T, H, W = frames.shape
        out = DeviceFrames.empty((T - 1, H, W), np.float32)
        _lib.check(_lib.load().lspiv_time_diff_dev(frames.c_ptr, frames.dtype_code, T, H, W, float(thres), int(bool(abs)), out.c_ptr, None))
        return out
    a = _lib.as_frames(frames)
    _lib.require_device()
    out = np.empty((a.shape[0] - 1,) + a.shape[1:], dtype=np.float32)
    _lib.check(_lib.load().lspiv_time_diff(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], a.shape[1], a.shape[2],
                                           float(thres), int(bool(abs)), _lib.ptr(out)))
    return out


def reduce_rolling(frames, samples: int = 25) -> np.ndarray:
    """``Frames.reduce_rolling`` on uint8 frames: trailing rolling mean of ``samples`` frames removed, clipped at 0,
    per-frame stretch to uint8 (the first ``samples - 1`` frames have no complete window and come out 0)."""
    a = frames if is_device(frames) else np.asarray(frames)
    if a.dtype != np.uint8 or a.ndim != 3:
        raise ValueError("reduce_rolling expects a (T, H, W) uint8 stack (grayscale camera frames)")
    if len(a) < samples:
        raise AssertionError(f"Amount of frames is smaller than requested rolling of {samples} samples")
    if is_device(a):
        out = DeviceFrames.empty(a.shape, np.uint8)
        _lib.check(_lib.load().lspiv_reduce_rolling_dev(a.c_ptr, a.shape[0], a.shape[1], a.shape[2], int(samples), out.c_ptr, None))
        return out
    a = np.ascontiguousarray(a)
    _lib.require_device()
    out = np.empty_like(a)
    _lib.check(_lib.load().lspiv_reduce_rolling(_lib.ptr(a), a.shape[0], a.shape[1], a.shape[2], int(samples), _lib.ptr(out)))
    return out


def range(frames) -> np.ndarray:  # noqa: A001 -- the reference's method name
    """``Frames.range``: (T, H, W) -> (H, W) in the frames' own dtype, maximum minus minimum through time (NaN skipped)."""
    if is_device(frames):
        out = np.empty(frames.shape[1:], dtype=frames.dtype)
        d_out = DeviceFrames.empty((1,) + frames.shape[1:], frames.dtype)
        _lib.check(_lib.load().lspiv_time_range_dev(frames.c_ptr, frames.dtype_code, *frames.shape, d_out.c_ptr, None))
        return d_out.to_host()[0]
    a = _lib.as_frames(frames)
    _lib.require_device()
    out = np.empty(a.shape[1:], dtype=a.dtype)
    _lib.check(_lib.load().lspiv_time_range(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], a.shape[1], a.shape[2], _lib.ptr(out)))
    return out


def minmax(frames, min=-np.inf, max=np.inf) -> np.ndarray:
    """``Frames.minmax`` on float32 frames: ``np.maximum(np.minimum(x, max), min)`` (NaN propagates)."""
    if is_device(frames):
        if frames.dtype != np.float32:
            raise ValueError("minmax on a device stack expects float32 frames (the output of edge_detect / smooth / time_diff)")
        out = DeviceFrames.empty(frames.shape, np.float32)
        _lib.check(_lib.load().lspiv_minmax_dev(frames.c_ptr, int(np.prod(frames.shape)), float(min), float(max), out.c_ptr, None))
        return out
    a = np.ascontiguousarray(frames, dtype=np.float32)
    _lib.require_device()
    out = np.empty_like(a)
    _lib.check(_lib.load().lspiv_minmax(_lib.ptr(a), a.size, float(min), float(max), _lib.ptr(out)))
    return out


def normalize(frames, samples: int = 15) -> np.ndarray:
    """``Frames.normalize`` on uint8 frames: sampled temporal mean removed, per-frame stretch to uint8."""
    a = frames if is_device(frames) else np.asarray(frames)
    if a.dtype != np.uint8 or a.ndim != 3:
        raise ValueError("normalize expects a (T, H, W) uint8 stack (grayscale camera frames)")
    if round(len(a) / samples) == 0:
        raise AssertionError(f"Amount of frames is too small to provide {samples} samples")
    if is_device(a):
        out = DeviceFrames.empty(a.shape, np.uint8)
        _lib.check(_lib.load().lspiv_normalize_dev(a.c_ptr, a.shape[0], a.shape[1], a.shape[2], int(samples), out.c_ptr, None))
        return out
    a = np.ascontiguousarray(a)
    _lib.require_device()
    out = np.empty_like(a)
    _lib.check(_lib.load().lspiv_normalize(_lib.ptr(a), a.shape[0], a.shape[1], a.shape[2], int(samples), _lib.ptr(out)))
    return out


def _blur(frames, k1: int, k2: int) -> np.ndarray:
    if is_device(frames):
        T, H, W = frames.shape
        out = DeviceFrames.empty(frames.shape, np.float32)
        lib = _lib.load()
        if k2:
            rc = lib.lspiv_edge_detect_dev(frames.c_ptr, frames.dtype_code, T, H, W, k1, k2, out.c_ptr, None)
        else:
            rc = lib.lspiv_gaussian_blur_dev(frames.c_ptr, frames.dtype_code, T, H, W, k1, out.c_ptr, None)
        _lib.check(rc)
        return out
    a = np.asarray(frames)
    single = a.ndim == 2
    a = _lib.as_frames(a[None] if single else a)
    _lib.require_device()
    out = np.empty(a.shape, dtype=np.float32)
    lib = _lib.load()
    if k2:
        rc = lib.lspiv_edge_detect(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], a.shape[1], a.shape[2], k1, k2, _lib.ptr(out))
    else:
        rc = lib.lspiv_gaussian_blur(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], a.shape[1], a.shape[2], k1, _lib.ptr(out))
    _lib.check(rc)
    return out[0] if single else out


def smooth(frames, wdw: int = 1) -> np.ndarray:
    """``Frames.smooth``: Gaussian blur with a (2 wdw + 1)^2 kernel, float32."""
    return _blur(frames, 2 * int(wdw) + 1, 0)


def edge_detect(frames, wdw_1: int = 1, wdw_2: int = 2) -> np.ndarray:
    """``Frames.edge_detect``: blur(2 wdw_2 + 1) - blur(2 wdw_1 + 1), float32."""
    return _blur(frames, 2 * int(wdw_1) + 1, 2 * int(wdw_2) + 1)
