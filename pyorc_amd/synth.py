"""Synthetic orthorectified particle-image stacks -- host generator.

Seeded particle images with a known, spatially varying flow, used by the parity tests and as
the small-size twin of the on-device generator (``lspiv_synth_*`` in the C ABI) that bench.py
uses for the 1000-pair stacks.  Nothing here is on the PIV path.

Flow (pixels / frame):  u(x, y) = 3.0 + 2.0 sin(2 pi y / H),  v(x, y) = 1.5 cos(2 pi x / W)
Particles: N_p = density * H * W Gaussian blobs, sigma = 1.2 px, peak intensity U(120, 255),
advected every frame and re-seeded at a random position when they leave the frame.
"""

from __future__ import annotations

import numpy as np

SIGMA = 1.2
RADIUS = 3  # 7x7 footprint


def flow_field(H: int, W: int, ys: np.ndarray, xs: np.ndarray):
    """Ground-truth displacement (u, v) in px/frame at positions (ys, xs)."""
    u = 3.0 + 2.0 * np.sin(2.0 * np.pi * ys / H)
    v = 1.5 * np.cos(2.0 * np.pi * xs / W)
    return u, v


def _render(H, W, py, px, amp):
    iy = np.rint(py).astype(np.int64)
    ix = np.rint(px).astype(np.int64)
    off = np.arange(-RADIUS, RADIUS + 1)
    yy = iy[:, None, None] + off[None, :, None]
    xx = ix[:, None, None] + off[None, None, :]
    w = amp[:, None, None] * np.exp(-((yy - py[:, None, None]) ** 2 + (xx - px[:, None, None]) ** 2) / (2 * SIGMA**2))
    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    ok = np.broadcast_to(ok, w.shape)
    idx = (yy * W + xx)
    idx = np.broadcast_to(idx, w.shape)[ok]
    img = np.bincount(idx, weights=w[ok], minlength=H * W).reshape(H, W)
    return img


def particle_stack(T: int, H: int, W: int, seed: int = 20260927, density: float = 0.02,
                   dtype=np.uint8, uniform_shift=None) -> np.ndarray:
    """(T, H, W) stack.  dtype uint8 (primary) or float32 (= uint8 minus temporal mean, signed).

    ``uniform_shift=(dx, dy)`` replaces the sinusoidal flow by a constant displacement.
    """
    rng = np.random.default_rng(seed)
    n_p = max(int(density * H * W), 1)
    py = rng.uniform(0, H, n_p)
    px = rng.uniform(0, W, n_p)
    amp = rng.uniform(120.0, 255.0, n_p)
    out = np.empty((T, H, W), dtype=np.uint8)
    for t in range(T):
        img = _render(H, W, py, px, amp)
        out[t] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if uniform_shift is None:
            u, v = flow_field(H, W, py, px)
        else:
            u, v = float(uniform_shift[0]), float(uniform_shift[1])
        px = px + u
        py = py + v
        gone = (px < -RADIUS) | (px >= W + RADIUS) | (py < -RADIUS) | (py >= H + RADIUS)
        k = int(gone.sum())
        if k:
            py[gone] = rng.uniform(0, H, k)
            px[gone] = rng.uniform(0, W, k)
            amp[gone] = rng.uniform(120.0, 255.0, k)
    if np.dtype(dtype) == np.uint8:
        return out
    f = out.astype(np.float32)
    f -= f.mean(axis=0, keepdims=True)
    return f.astype(dtype)


def projection_maps(src_shape, dst_shape, tilt: float = 0.35, seed: int = 0):
    """Index maps with the structure of CameraConfig.map_idx_img_ortho / map_mean_idx_img_ortho
    (pyorc/api/cameraconfig.py:739-860) for a synthetic pin-hole view: a plane seen under perspective, so the near
    part of the ortho grid is oversampled (several camera pixels per cell -> group means) and the far part is
    undersampled (nearest neighbour only).  Returns (idx_img, idx_ortho_mask, src_idx, uidx, norm_idx).
    """
    Hc, Wc = src_shape
    Ho, Wo = dst_shape
    rng = np.random.default_rng(seed)
    # homography ortho (col, row, 1) -> camera (u, v, w): perspective foreshortening along rows
    sx, sy = Wc / Wo, Hc / Ho
    Hm = np.array([[0.9 * sx, 0.12 * sx, 0.03 * Wc], [0.02 * sy, 0.8 * sy, 0.08 * Hc], [0.0, tilt / Ho, 1.0]])
    Hm[:2] *= 1.0 + tilt * 0.5
    Hm[0, 2] += rng.uniform(-2, 2)
    cols, rows = np.meshgrid(np.arange(Wo), np.arange(Ho))
    p = Hm @ np.stack([cols.ravel(), rows.ravel(), np.ones(Wo * Ho)])
    pc = np.int64(np.round(p[:2] / p[2]))
    inside = (pc[0] > 0) & (pc[0] < Wc) & (pc[1] > 0) & (pc[1] < Hc)            # cameraconfig.py:771-779
    idx_img = pc[1][inside] * Wc + pc[0][inside]                                # :791
    # camera pixels -> ortho cells (the inverse map), groups with more than one sample
    Hi = np.linalg.inv(Hm)
    coli, rowi = np.meshgrid(np.arange(Wc), np.arange(Hc))
    q = Hi @ np.stack([coli.ravel(), rowi.ravel(), np.ones(Wc * Hc)])
    ix = np.int64(np.floor(q[0] / q[2] + 0.5))
    iy = np.int64(np.floor(q[1] / q[2] + 0.5))
    ok = (iy >= 0) & (iy < Ho) & (ix >= 0) & (ix < Wo)                          # :829
    idx = iy[ok] * Wo + ix[ok]                                                  # :833
    src = (rowi.ravel()[ok] * Wc + coli.ravel()[ok])
    u, counts = np.unique(idx, return_counts=True)                              # :836
    keep = np.isin(idx, u[counts > 1])                                          # :839-841
    src_idx = src[keep]                                                         # :844
    uidx, norm_idx = np.unique(idx[keep], return_inverse=True)                  # :851
    return idx_img, inside, src_idx, uidx, norm_idx
