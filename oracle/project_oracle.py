"""CPU oracle of the orthoprojection step (N1 of SURVEY.md section 8f).  TEST INFRASTRUCTURE ONLY.

Unlike the PIV arithmetic, this algorithm IS in the reference tree, so it is restated line by line:
  pyorc/project.py:19-53    _group_average   (numba: sequential float32 sums per group, in sample order)
  pyorc/project.py:123-161  img_to_ortho     (nearest-neighbour scatter, then group means over the oversampled pixels)
  pyorc/project.py:164-230  project_numpy    (img_to_ortho applied to every frame)
  pyorc/api/frames.py:265   .fillna(0.0)     (missing values -> 0)
The reference module itself cannot be imported here (cv2, numba, xarray absent) and its tests pin shapes only
(tests/test_frames.py:29-50), so there are no reference vectors: parity is "restated from in-tree source, unpinned".
The index maps (CameraConfig.map_idx_img_ortho / map_mean_idx_img_ortho, api/cameraconfig.py:739-860) are INPUTS.
"""

from __future__ import annotations

import numpy as np


def group_average(data: np.ndarray, idx: np.ndarray, num_groups: int) -> np.ndarray:
    """pyorc/project.py:19-53: float32 sums accumulated in sample order, int64 counts, float32 averages."""
    data = np.asarray(data, dtype=np.float32)
    idx = np.asarray(idx, dtype=np.int64)
    order = np.argsort(idx, kind="stable")          # samples of a group keep their original order
    sidx = idx[order]
    counts = np.bincount(idx, minlength=num_groups).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    sums = np.zeros(num_groups, dtype=np.float32)
    sdata = data[order]
    for k in range(int(counts.max()) if len(counts) else 0):   # k-th sample of every group, sequential in k
        has = counts > k
        sums[has] = sums[has] + sdata[starts[has] + k]          # float32 + float32
    with np.errstate(all="ignore"):
        # numba: float32 / int64 -> float64 division stored to float32 (== correctly rounded float32 division)
        return (sums.astype(np.float64) / counts).astype(np.float32)


def img_to_ortho(img, shape, idx_img, idx_ortho, src_idx=None, uidx=None, norm_idx=None) -> np.ndarray:
    """pyorc/project.py:123-161; ``shape`` = (len(y), len(x)); ``idx_ortho`` boolean mask or flat indices."""
    img = np.float32(np.asarray(img).flatten())
    new_arr = np.zeros(shape[0] * shape[1])
    new_arr[idx_ortho] = img[idx_img]
    if src_idx is not None:
        samples_for_mean = img[src_idx]
        averages = group_average(samples_for_mean, norm_idx, len(uidx))
        new_arr[uidx] = averages
    return new_arr.reshape(shape[0], -1)


def project_frames(frames, shape, idx_img, idx_ortho, src_idx=None, uidx=None, norm_idx=None) -> np.ndarray:
    """project_numpy + Frames.project's fillna(0.0): (T, Hc, Wc) -> (T, Ho, Wo) float64 holding float32 values."""
    out = np.stack([img_to_ortho(f, shape, idx_img, idx_ortho, src_idx, uidx, norm_idx) for f in frames])
    return np.nan_to_num(out, nan=0.0, posinf=np.inf, neginf=-np.inf)
