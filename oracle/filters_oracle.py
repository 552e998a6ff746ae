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

import numpy as np


def normalize(frames: np.ndarray, samples: int = 15) -> np.ndarray:
    """frames.py:296-306.  uint8 frames: the sampled mean is exact (integer sums); result uint8."""
    frames = np.asarray(frames)
    time_interval = round(len(frames) / samples)
    assert time_interval != 0, f"Amount of frames is too small to provide {samples} samples"
    mean = frames[::time_interval].mean(axis=0).astype("float32")
    frames_reduce = frames.astype("float32") - mean
    frames_min = frames_reduce.min(axis=-1).min(axis=-1)[:, None, None]
    frames_max = frames_reduce.max(axis=-1).max(axis=-1)[:, None, None]
    with np.errstate(all="ignore"):
        q = (frames_reduce - frames_min) / (frames_max - frames_min) * 255
        q = np.where(np.isnan(q), np.float32(0), q)  # 0/0 for a constant frame: x86 casts NaN to 0
    return q.astype("uint8")


def minmax(frames: np.ndarray, min=-np.inf, max=np.inf) -> np.ndarray:
    """frames.py:362 on float32 frames (the dtype edge_detect / time_diff hand over)."""
    return np.maximum(np.minimum(np.asarray(frames, dtype=np.float32), np.float32(max)), np.float32(min))


def time_diff(frames: np.ndarray, thres: float = 0.0, abs: bool = False) -> np.ndarray:
    """frames.py:430-436: (T-1, H, W) float32."""
    d = np.diff(np.asarray(frames).astype(np.float32), axis=0)
    with np.errstate(invalid="ignore"):
        d = np.where(d > np.float32(thres), d, np.float32(0.0))  # .where(d > thres) -> NaN, then .fillna(0.0)
    return np.abs(d) if abs else d
