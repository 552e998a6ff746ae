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


# ----------------------------------------------------------------------------------------------
# project_cv  (pyorc/project.py:56-120): cv2.undistort + cv2.warpPerspective(flags=INTER_AREA)
# ----------------------------------------------------------------------------------------------
# OpenCV is a wheel dependency of the reference (pyproject.toml) and cannot be installed here, so -- like the Gaussian
# filters (oracle/filters_oracle.py) -- its PUBLISHED algorithm is restated; "unpinned against a real cv2".
#   cv2.undistort(img, K, dist)            = remap(img, initUndistortRectifyMap(K, dist, I, K, size, CV_16SC2),
#                                                  INTER_LINEAR, BORDER_CONSTANT 0)          (pyorc/cv.py:1392-1413)
#   cv2.warpPerspective(img, M, (w, h), flags=INTER_AREA) : INTER_AREA is replaced by INTER_LINEAR inside
#                                            warpPerspective; M maps source to destination and is inverted; border 0
#                                                                                           (pyorc/cv.py:993-1013)
# Both go through the same fixed-point remap: source coordinates are quantised to 1/32 pixel (INTER_BITS = 5,
# saturate_cast<int> = round half to even), 8-bit images are blended with integer weights that sum to 2^15 and
# rounded with + 2^14 >> 15, float images with the float weight table; neighbours outside the image count as 0.
# The intermediate (undistorted) image has the dtype of the input, exactly as in the reference's two-step chain.
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
WARP_BLOCK_W = 64   # WarpPerspectiveInvoker walks 64-pixel column blocks: X0 = M0 * x_block + ..., then X0 + M0 * x1


def _inv3(m) -> np.ndarray:
    """cv::invert on a 3 x 3 double matrix (DECOMP_LU takes the closed-form path for sizes <= 3): adjugate / determinant,
    evaluated in this order -- LAPACK's LU (np.linalg.inv) differs in the last bit, which moves 1/32-pixel rounding ties."""
    a, b, c, d, e, f, g, h, i = (float(v) for v in np.asarray(m, dtype=np.float64).reshape(9))
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    r = 1.0 / det
    return np.array([[(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r],
                     [(f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r],
                     [(d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r]])


def _round_half_even_to_int(a: np.ndarray) -> np.ndarray:
    a = np.clip(np.nan_to_num(a, nan=0.0, posinf=2.0**31 - 1, neginf=-2.0**31), -2.0**31, 2.0**31 - 1)
    return np.rint(a).astype(np.int64)


def undistort_map(camera_matrix, dist_coeffs, shape):
    """initUndistortRectifyMap(K, dist, R = I, newK = K, size) -> (ix, iy, frac) of every destination pixel:
    integer source coordinates and the 1/32-pixel fraction index fy * 32 + fx."""
    K = np.asarray(camera_matrix, dtype=np.float64).reshape(3, 3)
    d = np.zeros(8)
    dc = np.asarray(dist_coeffs, dtype=np.float64).ravel()
    if dc.size not in (0, 4, 5, 8):
        raise ValueError("dist_coeffs must hold 0, 4, 5 or 8 values (k1 k2 p1 p2 [k3 [k4 k5 k6]])")
    d[:dc.size] = dc
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ir = _inv3(K)
    H, W = shape
    i = np.arange(H, dtype=np.float64)[:, None]
    # the row walk of OpenCV: _x = i * ir[1] + ir[2], then _x += ir[0] per column (sequential double additions)
    def walk(a, b, c):
        start = i * a + b
        steps = np.full((H, W), c)
        steps[:, 0:1] = start
        return np.cumsum(steps, axis=1)
    _x, _y, _w = walk(ir[0, 1], ir[0, 2], ir[0, 0]), walk(ir[1, 1], ir[1, 2], ir[1, 0]), walk(ir[2, 1], ir[2, 2], ir[2, 0])
    w = 1.0 / _w
    x, y = _x * w, _y * w
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    u = fx * xd + u0
    v = fy * yd + v0
    iu = _round_half_even_to_int(u * INTER_TAB_SIZE)
    iv = _round_half_even_to_int(v * INTER_TAB_SIZE)
    return iu >> INTER_BITS, iv >> INTER_BITS, (iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1))


def warp_map(M, dst_shape):
    """cv2.warpPerspective's destination -> source map for the source-to-destination homography M."""
    Mi = _inv3(M).ravel()
    H, W = dst_shape
    xs = np.arange(W)
    xb = (xs // WARP_BLOCK_W) * WARP_BLOCK_W            # block start
    x1 = (xs - xb).astype(np.float64)
    xb = xb.astype(np.float64)
    y = np.arange(H, dtype=np.float64)[:, None]
    X0 = Mi[0] * xb + Mi[1] * y + Mi[2]
    Y0 = Mi[3] * xb + Mi[4] * y + Mi[5]
    W0 = Mi[6] * xb + Mi[7] * y + Mi[8]
    Wd = W0 + Mi[6] * x1
    with np.errstate(all="ignore"):
        Wd = np.where(Wd != 0, INTER_TAB_SIZE / Wd, 0.0)
        fX = np.clip((X0 + Mi[0] * x1) * Wd, -2.0**31, 2.0**31 - 1)
        fY = np.clip((Y0 + Mi[3] * x1) * Wd, -2.0**31, 2.0**31 - 1)
    X = _round_half_even_to_int(fX)
    Y = _round_half_even_to_int(fY)
    return X >> INTER_BITS, Y >> INTER_BITS, (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))


def remap_linear(img: np.ndarray, ix, iy, frac) -> np.ndarray:
    """cv2.remap(img, (ix, iy), frac, INTER_LINEAR, BORDER_CONSTANT 0) on one uint8 or float32 frame."""
    img = np.asarray(img)
    Hs, Ws = img.shape
    # OpenCV stores the integer coordinates as short (saturated): anything that saturates is outside any real image
    ix = np.clip(ix, -32768, 32767)
    iy = np.clip(iy, -32768, 32767)
    fx = (frac & (INTER_TAB_SIZE - 1)).astype(np.int64)
    fy = (frac >> INTER_BITS).astype(np.int64)

    def px(yy, xx):
        ok = (yy >= 0) & (yy < Hs) & (xx >= 0) & (xx < Ws)
        return np.where(ok, img[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)], 0)

    p00, p01, p10, p11 = px(iy, ix), px(iy, ix + 1), px(iy + 1, ix), px(iy + 1, ix + 1)
    if img.dtype == np.uint8:
        w00 = (32 - fx) * (32 - fy) * 32          # = round(((1 - a)(1 - b)) * 2^15), exact
        w01 = fx * (32 - fy) * 32
        w10 = (32 - fx) * fy * 32
        w11 = fx * fy * 32
        acc = p00.astype(np.int64) * w00 + p01.astype(np.int64) * w01 + p10.astype(np.int64) * w10 + p11.astype(np.int64) * w11
        return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    a, b = fx.astype(np.float32) / np.float32(32), fy.astype(np.float32) / np.float32(32)
    one = np.float32(1)
    w00, w01, w10, w11 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b      # float32 table entries
    f = lambda p: p.astype(np.float32)
    return ((f(p00) * w00 + f(p01) * w01) + f(p10) * w10) + f(p11) * w11                  # left to right, float32


def project_cv(frames, camera_matrix, dist_coeffs, M, dst_shape) -> np.ndarray:
    """pyorc/project.py:56-120 on a (T, Hc, Wc) uint8 or float32 stack -> (T, Ho, Wo) of the same dtype."""
    frames = np.asarray(frames)
    if frames.dtype not in (np.uint8, np.float32):
        frames = frames.astype(np.float32)
    m1 = undistort_map(camera_matrix, dist_coeffs, frames.shape[1:])
    m2 = warp_map(M, dst_shape)
    return np.stack([remap_linear(remap_linear(f, *m1), *m2) for f in frames])
