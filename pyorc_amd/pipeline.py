"""Device-resident chain of the stages around the hot path: camera frames -> velocities, one H2D, one D2H.

In pyorc the stages run one after the other on the host, each materialising a full stack (the Ngwerere recipe,
examples/ngwerere/ngwerere.yml:5-18): ``frames.normalize()`` (uint8) -> [edge filter] -> ``frames.project()`` (float64,
x4..x8 the bytes of the camera stack) -> ``get_piv()`` (window stack x3.9, correlation volume 32 MB/pair) ->
``to_netcdf`` (int16).  ``CameraToVelocity`` keeps everything between the raw uint8 camera frames and the result
block in HBM and calls only ``*_dev`` entry points of the C ABI:

    H2D uint8 frames -> lspiv_normalize_dev -> lspiv_edge_detect_dev -> lspiv_minmax_dev   (each optional)
                     -> lspiv_project_frames_dev -> lspiv_piv_pairs_dev
                     -> lspiv_pack_int16_dev (optional) -> D2H (16 B or 8 B per vector)

Every stage is the same kernel the stand-alone mirrors (``filters``, ``project``, ``piv``) call, so the chain is
bit-identical to running them one by one (tested).  The camera-geometry index maps are pyorc's
(``CameraConfig.map_idx_img_ortho`` / ``map_mean_idx_img_ortho``).  With ``normalize_samples=15, edge_detect=(1, 2),
minmax=(-5, 5)`` the chain is the whole ``frames:`` + ``get_piv`` part of the Ngwerere recipe.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib, window
from .project import Projection


class _DevBuf:
    """Grow-only HBM buffer owned by the pipeline (lspiv_dev_malloc / lspiv_dev_free)."""

    def __init__(self):
        self.ptr = C.c_void_p()
        self.cap = 0

    def ensure(self, nbytes: int) -> C.c_void_p:
        if nbytes > self.cap:
            self.free()
            _lib.check(_lib.load().lspiv_dev_malloc(C.byref(self.ptr), nbytes))
            self.cap = nbytes
        return self.ptr

    def free(self):
        if self.ptr:
            _lib.load().lspiv_dev_free(self.ptr)
        self.ptr, self.cap = C.c_void_p(), 0


class CameraToVelocity:
    """uint8 camera frames (T, Hc, Wc) -> u, v [px], corr_max, s2n (T-1, n_rows, n_cols), all stages on the GPU."""

    def __init__(self, cam_shape, ortho_shape, idx_img, idx_ortho, src_idx=None, uidx=None, norm_idx=None,
                 window_size=(32, 32), overlap=(16, 16), normalize_samples: Optional[int] = None,
                 signal_threshold: Optional[float] = None, edge_detect: Optional[tuple] = None,
                 minmax: Optional[tuple] = None):
        _lib.require_device()
        self.cam_shape = (int(cam_shape[0]), int(cam_shape[1]))
        self.ortho_shape = (int(ortho_shape[0]), int(ortho_shape[1]))
        self.window_size, self.overlap = tuple(window_size), tuple(overlap)
        self.normalize_samples = normalize_samples
        self.signal_threshold = -1.0 if signal_threshold is None else float(signal_threshold)
        self.edge_detect = None if edge_detect is None else (2 * int(edge_detect[0]) + 1, 2 * int(edge_detect[1]) + 1)
        self.minmax = None if minmax is None else (float("-inf") if minmax[0] is None else float(minmax[0]),
                                                   float("inf") if minmax[1] is None else float(minmax[1]))
        self.n_rows, self.n_cols = window.get_array_shape(self.ortho_shape, self.window_size, self.overlap)
        if self.n_rows < 1 or self.n_cols < 1:
            raise ValueError("ortho frame smaller than the interrogation window")
        self.projection = Projection(self.cam_shape, self.ortho_shape, idx_img, idx_ortho, src_idx, uidx, norm_idx)
        self._cam, self._norm, self._edge, self._ortho, self._out, self._packed = (_DevBuf() for _ in range(6))

    def run(self, frames, packed: bool = False):
        """Returns (u, v, corr_max, s2n) float32, or their int16 packing (scale 0.01, fill -9999) when ``packed``.

        Note ``packed`` encodes the PIXEL displacements; pyorc packs velocities in m/s -- scale by res/dt on the host
        first (``velocimetry.get_ffpiv``) when that is what goes to disk.
        """
        a = np.ascontiguousarray(frames)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[1:] != self.cam_shape:
            raise ValueError(f"expected a (T, {self.cam_shape[0]}, {self.cam_shape[1]}) uint8 stack, got {a.shape} {a.dtype}")
        T = a.shape[0]
        if T < 2:
            raise ValueError("need at least two frames")
        lib = _lib.load()
        n_cam = self.cam_shape[0] * self.cam_shape[1]
        n_ortho = self.ortho_shape[0] * self.ortho_shape[1]
        n_vec = (T - 1) * self.n_rows * self.n_cols
        d_cam = self._cam.ensure(T * n_cam)
        _lib.check(lib.lspiv_memcpy_h2d(d_cam, _lib.ptr(a), a.nbytes))
        src = d_cam
        if self.normalize_samples:
            if round(T / self.normalize_samples) == 0:
                raise AssertionError(f"Amount of frames is too small to provide {self.normalize_samples} samples")
            src = self._norm.ensure(T * n_cam)
            _lib.check(lib.lspiv_normalize_dev(d_cam, T, self.cam_shape[0], self.cam_shape[1], self.normalize_samples, src, None))
        src_dtype = np.uint8
        if self.edge_detect:
            d_edge = self._edge.ensure(T * n_cam * 4)
            _lib.check(lib.lspiv_edge_detect_dev(src, 0, T, self.cam_shape[0], self.cam_shape[1], self.edge_detect[0],
                                                 self.edge_detect[1], d_edge, None))
            src, src_dtype = d_edge, np.float32
        if self.minmax:
            if src_dtype is np.uint8:
                raise ValueError("minmax in the chain follows edge_detect (float32 frames); uint8 frames are not thresholded")
            _lib.check(lib.lspiv_minmax_dev(src, T * n_cam, self.minmax[0], self.minmax[1], src, None))   # in place
        d_ortho = self._ortho.ensure(T * n_ortho * 4)
        self.projection.project_frames_dev(src.value, src_dtype, T, d_ortho.value)
        d_out = self._out.ensure(4 * n_vec * 4)
        _lib.check(lib.lspiv_piv_pairs_dev(d_ortho, 1, T, self.ortho_shape[0], self.ortho_shape[1], self.window_size[0],
                                           self.window_size[1], self.overlap[0], self.overlap[1], self.signal_threshold,
                                           d_out, None, None))
        shape = (4, T - 1, self.n_rows, self.n_cols)
        if packed:
            d_pk = self._packed.ensure(4 * n_vec * 2)
            _lib.check(lib.lspiv_pack_int16_dev(d_out, 4 * n_vec, 0.01, -9999, d_pk, None))
            res = np.empty(shape, dtype=np.int16)
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(res), d_pk, res.nbytes))
        else:
            res = np.empty(shape, dtype=np.float32)
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(res), d_out, res.nbytes))
        return res[0], res[1], res[2], res[3]

    def close(self):
        self.projection.close()
        for b in (self._cam, self._norm, self._edge, self._ortho, self._out, self._packed):
            b.free()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
