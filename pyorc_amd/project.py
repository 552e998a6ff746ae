"""Mirror of ``pyorc/project.py`` (``img_to_ortho`` / ``project_numpy`` and ``project_cv``) on the MI355X -- SURVEY.md section 8f row N1.

The camera-geometry part stays in pyorc: ``CameraConfig.map_idx_img_ortho`` and ``map_mean_idx_img_ortho``
(pyorc/api/cameraconfig.py:739-860) produce the index maps; this module consumes them.  ``Projection`` uploads the
maps once; ``project_frames`` then turns a (T, Hc, Wc) stack of camera frames into the (T, Ho, Wo) float32 stack
the PIV engine reads -- one gather kernel per call, bit-identical to the reference's numba loop, NaN -> 0 like
``Frames.project`` (pyorc/api/frames.py:265).  With ``device=True`` the projected stack stays in HBM and is handed
to ``lspiv_piv_pairs_dev`` directly: the only host->device traffic is then the raw uint8 camera frames.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .device import DeviceFrames, is_device


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a).ravel(), dtype=np.int64)


class Projection:
    """Device-resident projection plan for one camera configuration / water level."""

    def __init__(self, src_shape, dst_shape, idx_img, idx_ortho, src_idx=None, uidx=None, norm_idx=None):
        lib = _lib.load()
        _lib.require_device()
        self.src_shape = (int(src_shape[0]), int(src_shape[1]))
        self.dst_shape = (int(dst_shape[0]), int(dst_shape[1]))
        idx_ortho = np.asarray(idx_ortho)
        if idx_ortho.dtype == np.bool_:  # the reference passes a mask: new_arr[mask] = img[idx_img] (project.py:154)
            idx_ortho = np.flatnonzero(idx_ortho.ravel())
        ii, io = _i64(idx_img), _i64(idx_ortho)
        if ii.shape != io.shape:
            raise ValueError(f"idx_img has {ii.size} entries, idx_ortho selects {io.size}")
        if src_idx is None:  # reducer != "mean": nearest neighbour only (project.py:196-199)
            si = ni = ui = np.zeros(0, np.int64)
        else:
            si, ni, ui = _i64(src_idx), _i64(norm_idx), _i64(uidx)
            if si.shape != ni.shape:
                raise ValueError("src_idx and norm_idx must have the same length")
        self.nearest_only = ui.size == 0   # no averaged cells: uint8 frames may stay uint8 (project_frames, keep_uint8)
        self._h = C.c_void_p()
        _lib.check(lib.lspiv_projection_create(self.src_shape[0], self.src_shape[1], self.dst_shape[0], self.dst_shape[1],
                                               _lib.ptr(ii), _lib.ptr(io), ii.size, _lib.ptr(si), _lib.ptr(ni), si.size,
                                               _lib.ptr(ui), ui.size, C.byref(self._h)))

    def project_frames(self, frames, keep_uint8: Optional[bool] = None) -> np.ndarray:
        """(T, Hc, Wc) or (Hc, Wc) camera frames -> (T, Ho, Wo) float32 (``project_numpy`` + ``fillna(0)``).
        A ``DeviceFrames`` stack is projected in HBM and a ``DeviceFrames`` comes back.

        ``keep_uint8``: a nearest-neighbour-only plan (``reducer`` other than "mean", pyorc/project.py:196-199) gives every
        cell a source byte or 0, so a uint8 stack can stay uint8 -- the same values in a quarter of the bytes, and
        ``get_piv`` then runs its uint8 kernels.  Default: on for ``DeviceFrames`` (the stack stays in HBM for the next
        stage), off for host arrays (the reference hands out floats); ``ValueError`` when asked for on a plan with
        group means or on float frames."""
        dev = is_device(frames)
        dt = frames.dtype if dev else np.asarray(frames).dtype
        can = self.nearest_only and np.dtype(dt) == np.uint8
        if keep_uint8 and not can:
            raise ValueError("keep_uint8 needs uint8 frames and a plan without group means (reducer other than 'mean')")
        u8 = can and (dev if keep_uint8 is None else bool(keep_uint8))
        if dev:
            if frames.shape[1:] != self.src_shape:
                raise ValueError(f"frames are {frames.shape[1:]}, projection expects {self.src_shape}")
            out = DeviceFrames.empty((frames.shape[0],) + self.dst_shape, np.uint8 if u8 else np.float32)
            self.project_frames_dev(frames.ptr, frames.dtype, frames.shape[0], out.ptr, keep_uint8=u8)
            return out
        a = np.asarray(frames)
        single = a.ndim == 2
        a = _lib.as_frames(a[None] if single else a)
        if a.shape[1:] != self.src_shape:
            raise ValueError(f"frames are {a.shape[1:]}, projection expects {self.src_shape}")
        if u8:
            out = np.empty((a.shape[0],) + self.dst_shape, dtype=np.uint8)
            _lib.check(_lib.load().lspiv_project_frames_u8(self._h, _lib.ptr(a), a.shape[0], _lib.ptr(out)))
            return out[0] if single else out
        out = np.empty((a.shape[0],) + self.dst_shape, dtype=np.float32)
        _lib.check(_lib.load().lspiv_project_frames(self._h, _lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], _lib.ptr(out)))
        return out[0] if single else out

    def project_frames_dev(self, d_frames: int, dtype, T: int, d_out: int, stream: Optional[int] = None,
                           keep_uint8: bool = False) -> None:
        """Device pointers in, device pointer out (see bench / tools/project_bench.py); ``keep_uint8``: ``d_out`` is a uint8
        stack (uint8 frames, nearest-neighbour-only plan)."""
        if keep_uint8:
            if np.dtype(dtype) != np.uint8:
                raise ValueError("keep_uint8 needs uint8 frames")
            _lib.check(_lib.load().lspiv_project_frames_u8_dev(self._h, C.c_void_p(d_frames), T, C.c_void_p(d_out),
                                                               C.c_void_p(stream) if stream else None))
            return
        _lib.check(_lib.load().lspiv_project_frames_dev(self._h, C.c_void_p(d_frames), _lib.DTYPE_CODES[np.dtype(dtype)], T,
                                                        C.c_void_p(d_out), C.c_void_p(stream) if stream else None))

    def project_into(self, frames, out, f0: int = 0) -> None:
        """``DeviceFrames`` camera frames -> frames ``[f0, f0 + len(frames))`` of the HBM-resident ortho stack ``out`` (float32; uint8 for
        a nearest-neighbour-only plan fed with uint8): how ``pyorc_amd.resident`` fills the stack the PIV kernels read."""
        n = frames.shape[0]
        if frames.shape[1:] != self.src_shape or out.shape[1:] != self.dst_shape or not 0 <= f0 <= out.shape[0] - n:
            raise ValueError(f"{frames.shape} frames do not project into frames [{f0}, {f0 + n}) of a {out.shape} stack with this plan")
        if out.dtype not in (np.dtype(np.float32), np.dtype(np.uint8)):
            raise TypeError(f"ortho stacks are float32 (or uint8), got {out.dtype}")
        fb = self.dst_shape[0] * self.dst_shape[1] * out.dtype.itemsize
        self.project_frames_dev(frames.ptr, frames.dtype, n, out.ptr + int(f0) * fb, keep_uint8=out.dtype == np.uint8)

    def close(self):
        if self._h:
            _lib.load().lspiv_projection_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProjectionCV:
    """``Frames.project(method="cv")`` (pyorc/project.py:56-120): ``cv2.undistort`` + ``cv2.warpPerspective`` on the GPU.

    ``camera_matrix`` (3x3) and ``dist_coeffs`` (k1 k2 p1 p2 [k3 [k4 k5 k6]]) are ``CameraConfig.camera_matrix`` /
    ``.dist_coeffs``; ``M`` is the source-to-destination homography pyorc builds with ``cv.get_M_2D(src, dst)``
    (pyorc/cv.py:769-795) from the undistorted bounding-box corners; ``dst_shape`` = (len(y), len(x)).  The output keeps
    the dtype of the frames (uint8 / float32) like the reference.  OpenCV's fixed-point remap is restated, not pinned
    against a real cv2 (see include/lspiv.h)."""

    def __init__(self, src_shape, dst_shape, camera_matrix, dist_coeffs, M):
        lib = _lib.load()
        _lib.require_device()
        self.src_shape = (int(src_shape[0]), int(src_shape[1]))
        self.dst_shape = (int(dst_shape[0]), int(dst_shape[1]))
        K = None if camera_matrix is None else np.ascontiguousarray(np.asarray(camera_matrix, dtype=np.float64).reshape(9))
        d = np.zeros(0) if dist_coeffs is None else np.ascontiguousarray(np.asarray(dist_coeffs, dtype=np.float64).ravel())
        Mh = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(9))
        self._h = C.c_void_p()
        _lib.check(lib.lspiv_project_cv_create(self.src_shape[0], self.src_shape[1], self.dst_shape[0], self.dst_shape[1],
                                               _lib.ptr(K) if K is not None else None, _lib.ptr(d) if d.size else None, d.size,
                                               _lib.ptr(Mh), C.byref(self._h)))

    def project_frames(self, frames):
        """(T, Hc, Wc) or (Hc, Wc) uint8 / float32 frames -> (T, Ho, Wo) of the same dtype; DeviceFrames stay in HBM."""
        if is_device(frames):
            if frames.shape[1:] != self.src_shape or frames.dtype == np.float64:
                raise ValueError(f"expected uint8 / float32 frames of shape {self.src_shape}, got {frames.shape[1:]} {frames.dtype}")
            out = DeviceFrames.empty((frames.shape[0],) + self.dst_shape, frames.dtype)
            _lib.check(_lib.load().lspiv_project_cv_frames_dev(self._h, frames.c_ptr, frames.dtype_code, frames.shape[0], out.c_ptr, None))
            return out
        a = np.asarray(frames)
        single = a.ndim == 2
        a = a[None] if single else a
        if a.dtype != np.uint8:
            a = a.astype(np.float32)   # cv2 would keep float64; the kernels compute in float32 like every other stage
        a = np.ascontiguousarray(a)
        if a.shape[1:] != self.src_shape:
            raise ValueError(f"frames are {a.shape[1:]}, projection expects {self.src_shape}")
        out = np.empty((a.shape[0],) + self.dst_shape, dtype=a.dtype)
        _lib.check(_lib.load().lspiv_project_cv_frames(self._h, _lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0], _lib.ptr(out)))
        return out[0] if single else out

    def close(self):
        if self._h:
            _lib.load().lspiv_project_cv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def project_cv(frames, camera_matrix, dist_coeffs, M, dst_shape):
    """One-shot form of :class:`ProjectionCV` (the arithmetic of pyorc/project.py:56-120 on a frame stack)."""
    a = frames if is_device(frames) else np.asarray(frames)
    src_shape = a.shape[-2:]
    p = ProjectionCV(src_shape, dst_shape, camera_matrix, dist_coeffs, M)
    try:
        return p.project_frames(a)
    finally:
        p.close()


def img_to_ortho(img, x, y, idx_img, idx_ortho, src_idx=None, uidx=None, norm_idx=None) -> np.ndarray:
    """``pyorc.project.img_to_ortho`` drop-in (same arguments, project.py:123); returns float32 instead of float64."""
    img = np.asarray(img)
    p = Projection(img.shape, (len(y), len(x)), idx_img, idx_ortho, src_idx, uidx, norm_idx)
    try:
        return p.project_frames(img)
    finally:
        p.close()


def pack_int16(values, scale: float = 0.01, fill: int = -9999) -> np.ndarray:
    """On-device int16 packing of a result variable (pyorc/const.py:80-83), SURVEY.md section 8f row N4."""
    a = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.int16)
    _lib.require_device()
    _lib.check(_lib.load().lspiv_pack_int16(_lib.ptr(a), a.size, float(scale), int(fill), _lib.ptr(out)))
    return out
