"""CPU oracle of the element-wise pre-processing filters (N2 of SURVEY.md section 8f).  TEST INFRASTRUCTURE ONLY.

Restated line by line from the reference tree (these do not depend on ffpiv / cv2):
  Frames.normalize   pyorc/api/frames.py:279-306   temporal mean of sampled frames removed, per-frame stretch to uint8
  Frames.minmax      pyorc/api/frames.py:344-362   np.maximum(np.minimum(x, max), min)
  Frames.time_diff   pyorc/api/frames.py:409-436   float32 difference in time, values <= thres (and NaN) -> 0, optional abs
``edge_detect`` / ``smooth`` (frames.py:308-342,438-467) are cv2.GaussianBlur calls (pyorc/cv.py:142-183); cv2 is not
installable here, so they are NOT restated -- see DESIGN.md section 8.  The reference's tests pin shapes only
(tests/test_frames.py:55-110), so these restatements are "from in-tree source, unpinned".
"""

from __future__ import annotations

import warnings

import numpy as np


def normalize(frames: np.ndarray, samples: int = 15) -> np.ndarray:
    """frames.py:296-306.  uint8 frames: the sampled mean is exact (integer sums); result uint8."""
    frames = np.asarray(frames)
    time_interval = round(len(frames) / samples)
    assert time_interval != 0, f"Amount of frames is too small to provide {samples} samples"
    mean = frames[::time_interval].mean(axis=0).astype("float32")
    return normalize_with_mean(frames, mean)


def normalize_with_mean(frames: np.ndarray, mean: np.ndarray) -> np.ndarray:
    """The per-frame part of ``normalize`` (frames.py:300-306) for a given sampled mean -- tests of stacks too long for numpy check single frames with it."""
    frames_reduce = np.asarray(frames).astype("float32") - mean
    frames_min = frames_reduce.min(axis=-1).min(axis=-1)[:, None, None]
    frames_max = frames_reduce.max(axis=-1).max(axis=-1)[:, None, None]
    with np.errstate(all="ignore"):
        q = (frames_reduce - frames_min) / (frames_max - frames_min) * 255
        q = np.where(np.isnan(q), np.float32(0), q)  # 0/0 for a constant frame: x86 casts NaN to 0
    return q.astype("uint8")


def minmax(frames: np.ndarray, min=-np.inf, max=np.inf) -> np.ndarray:
    """frames.py:362 on float32 frames (the dtype edge_detect / time_diff hand over)."""
    return np.maximum(np.minimum(np.asarray(frames, dtype=np.float32), np.float32(max)), np.float32(min))


def reduce_rolling(frames: np.ndarray, samples: int = 25) -> np.ndarray:
    """frames.py:381-407 on uint8 frames, float64 like xarray: ``roll = rolling(time=samples).mean()`` (trailing window,
    NaN until it is complete; the window sum of uint8 samples is an exact integer, so sum / samples has one rounding
    whatever the summation order -- numpy's nanmean path; bottleneck's move_mean multiplies by 1 / samples instead and
    may differ in the last bit), ``thres = maximum(frames - roll, 0)``, ``(thres * 255 / thres.max over the frame)
    .astype(uint8).where(roll != 0, 0)``.  NaN -> uint8 (incomplete windows, 0 / 0 for a frame whose maximum is 0) is
    undefined in C; numpy on x86-64 gives 0, which is what is restated here.  UNPINNED against a real xarray run."""
    a = np.asarray(frames)
    assert len(a) >= samples, f"Amount of frames is smaller than requested rolling of {samples} samples"
    cs = np.cumsum(a.astype(np.int64), axis=0)
    win = cs[samples - 1:].copy()
    win[1:] -= cs[:-samples]
    roll = win / float(samples)                                                  # (T - samples + 1, H, W) float64
    x = a[samples - 1:].astype(np.float64)
    thres = np.maximum(x - roll, 0.0)
    fmax = thres.max(axis=-1).max(axis=-1)[:, None, None]
    with np.errstate(all="ignore"):
        q = thres * 255 / fmax
    q = np.where(np.isnan(q) | (roll == 0), 0.0, q)
    out = np.zeros(a.shape, np.uint8)
    out[samples - 1:] = q.astype(np.uint8)
    return out


def time_range(frames: np.ndarray) -> np.ndarray:
    """frames.py:364-379: (max(dim="time") - min(dim="time")).astype(dtype); xarray skips NaN for float frames."""
    a = np.asarray(frames)
    if a.dtype.kind == "f":
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)                       # all-NaN pixels -> NaN
            return (np.nanmax(a, axis=0) - np.nanmin(a, axis=0)).astype(a.dtype)
    return (a.max(axis=0) - a.min(axis=0)).astype(a.dtype)


def time_diff(frames: np.ndarray, thres: float = 0.0, abs: bool = False) -> np.ndarray:
    """frames.py:430-436: (T-1, H, W) float32."""
    d = np.diff(np.asarray(frames).astype(np.float32), axis=0)
    with np.errstate(invalid="ignore"):
        d = np.where(d > np.float32(thres), d, np.float32(0.0))  # .where(d > thres) -> NaN, then .fillna(0.0)
    return np.abs(d) if abs else d


# ---------------------------------------------------------------------------------------------------------------------
# cv2.GaussianBlur-based filters -- THIRD-PARTY ALGORITHM RESTATED, UNPINNED
# ---------------------------------------------------------------------------------------------------------------------
# Frames.smooth (pyorc/api/frames.py:438-467 -> pyorc/cv.py:142-158) and Frames.edge_detect (frames.py:308-342 ->
# cv.py:162-183) call cv2.GaussianBlur(img.astype("float32"), (k, k), 0).  OpenCV (opencv-python, unpinned in
# pyproject.toml) is not installable here, so its published algorithm is restated:
#   * kernel for sigma <= 0: fixed tables for k = 1, 3, 5, 7 ([1], [.25 .5 .25], [.0625 .25 .375 .25 .0625],
#     [.03125 .109375 .21875 .28125 ...]); otherwise sigma = 0.3*((k-1)*0.5 - 1) + 0.8 and exp(-x^2 / 2 sigma^2)
#     normalised to sum 1, coefficients stored as float32 (getGaussianKernel, ktype CV_32F);
#   * separable: rows first, then columns, float32 intermediates; borders BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba);
#   * symmetric evaluation k0*x0 + sum_j kj*(x[-j] + x[+j]).
# OpenCV's SIMD paths may fuse multiply-adds, so agreement with a real cv2 is expected to ~1e-6 relative, not bitwise.
_SMALL_TAB = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
              7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_kernel(ksize: int) -> np.ndarray:
    if ksize in _SMALL_TAB:
        return np.asarray(_SMALL_TAB[ksize], dtype=np.float32)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def gaussian_blur(img: np.ndarray, ksize: int) -> np.ndarray:
    """cv2.GaussianBlur(img.astype(float32), (ksize, ksize), 0) restated; img (H, W) -> float32 (H, W)."""
    a = np.asarray(img).astype(np.float32)
    k = gaussian_kernel(ksize)
    r = ksize // 2
    H, W = a.shape

    def pass1d(x, axis):
        n = x.shape[axis]
        idx = np.arange(n)
        out = np.take(x, idx, axis=axis) * k[r]
        for j in range(1, r + 1):
            lo = np.take(x, _reflect101(idx - j, n), axis=axis)
            hi = np.take(x, _reflect101(idx + j, n), axis=axis)
            out = out + k[r + j] * (lo + hi)          # float32 throughout
        return out.astype(np.float32)

    return pass1d(pass1d(a, 1), 0)


def smooth(frames: np.ndarray, wdw: int = 1) -> np.ndarray:
    """Frames.smooth: stride = 2*wdw + 1 (frames.py:455)."""
    return np.stack([gaussian_blur(f, 2 * wdw + 1) for f in np.asarray(frames)])


def edge_detect(frames: np.ndarray, wdw_1: int = 1, wdw_2: int = 2) -> np.ndarray:
    """Frames.edge_detect: blur(stride_2) - blur(stride_1) (cv.py:180-183)."""
    return np.stack([gaussian_blur(f, 2 * wdw_2 + 1) - gaussian_blur(f, 2 * wdw_1 + 1) for f in np.asarray(frames)])
