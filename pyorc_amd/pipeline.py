"""Device-resident chain of the stages around the hot path: camera frames -> velocities, one H2D, one D2H.

In pyorc the stages run one after the other on the host, each materialising a full stack (the Ngwerere recipe,
examples/ngwerere/ngwerere.yml:5-18): ``frames.normalize()`` (uint8) -> [edge filter] -> ``frames.project()`` (float64,
x4..x8 the bytes of the camera stack) -> ``get_piv()`` (window stack x3.9, correlation volume 32 MB/pair) ->
``to_netcdf`` (int16).  ``CameraToVelocity`` keeps everything between the raw uint8 camera frames and the result
block in HBM and calls only ``*_dev`` entry points of the C ABI:

    H2D uint8 frames -> lspiv_normalize_dev -> lspiv_edge_detect_clip_dev (edge_detect + minmax in one pass)   (each optional)
                     -> lspiv_project_frames_dev (lspiv_project_frames_u8_dev: nearest-neighbour-only plan, uint8 in) -> lspiv_piv_pairs_dev
                     -> lspiv_pack_int16_dev (optional) -> D2H (16 B or 8 B per vector)

Every stage is the same kernel the stand-alone mirrors (``filters``, ``project``, ``piv``) call, so the chain is
bit-identical to running them one by one (tested).

The upload is the longest single step of the chain (1080p: 415 MB per 200 frames at ~45 GB/s of PCIe against ~9 ms of
kernels), so ``run`` can stream (by default where the kernels weigh enough): the frames ``normalize`` samples go first (their mean plane is all the filter needs
of the rest of the stack), then the stack arrives in time chunks cut on multiples of ``lspiv_chunk_alignment`` pairs, and
while chunk k+1 crosses PCIe on the library's copy stream, chunk k runs normalise -> [edge] -> project -> PIV on a compute
stream of the chain's own; the result block of chunk k-1 goes back in between.  Every stage is per-frame (or, the PIV,
anchored to the absolute pair index), so the streamed run returns the bits of the one-piece run (tested).  The camera-geometry index maps are pyorc's
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
        if self.minmax and not self.edge_detect:   # checked before any buffer exists
            raise ValueError("minmax in the chain follows edge_detect (float32 frames); uint8 frames are not thresholded")
        self.n_rows, self.n_cols = window.get_array_shape(self.ortho_shape, self.window_size, self.overlap)
        if self.n_rows < 1 or self.n_cols < 1:
            raise ValueError("ortho frame smaller than the interrogation window")
        self.projection = Projection(self.cam_shape, self.ortho_shape, idx_img, idx_ortho, src_idx, uidx, norm_idx)
        # a nearest-neighbour-only plan fed with uint8 frames (no edge filter before it): the ortho stack stays uint8 and the
        # PIV runs its uint8 kernels on the same values (Projection.project_frames, keep_uint8)
        self.ortho_uint8 = self.projection.nearest_only and self.edge_detect is None
        self._cam, self._norm, self._edge, self._ortho, self._out, self._packed, self._mean = (_DevBuf() for _ in range(7))
        self._comp = None                                   # compute stream of the streamed run

    def run(self, frames, packed: bool = False, streamed: Optional[bool] = None, n_chunks: int = 8):
        """Returns (u, v, corr_max, s2n) float32, or their int16 packing (scale 0.01, fill -9999) when ``packed``.

        ``streamed`` (default: stacks of 64 MiB and more with windows above 32 px or the edge filter): upload in ``n_chunks`` time chunks overlapped with the kernels of
        the previous chunk; same bits as the one-piece run.  Note ``packed`` encodes the PIXEL displacements; pyorc packs
        velocities in m/s -- scale by res/dt on the host first (``velocimetry.get_ffpiv``) when that is what goes to disk.
        """
        a = np.ascontiguousarray(frames)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[1:] != self.cam_shape:
            raise ValueError(f"expected a (T, {self.cam_shape[0]}, {self.cam_shape[1]}) uint8 stack, got {a.shape} {a.dtype}")
        T = a.shape[0]
        if T < 2:
            raise ValueError("need at least two frames")
        if self.normalize_samples and round(T / self.normalize_samples) == 0:
            raise AssertionError(f"Amount of frames is too small to provide {self.normalize_samples} samples")
        if streamed is None:
            # worth it where the kernels weigh about as much as the upload (measured): windows above
            # 32 px or the edge filter; at 32 x 32 the chain is 80 % upload and the chunking only costs
            streamed = a.nbytes >= (64 << 20) and (min(self.window_size) > 32 or self.edge_detect is not None)
        if streamed:
            bounds = self._chunk_bounds(T - 1, n_chunks)
            if len(bounds) > 2:
                return self._run_streamed(a, bounds, packed)
        lib = _lib.load()
        n_cam = self.cam_shape[0] * self.cam_shape[1]
        n_ortho = self.ortho_shape[0] * self.ortho_shape[1]
        n_vec = (T - 1) * self.n_rows * self.n_cols
        d_cam = self._cam.ensure(T * n_cam)
        _lib.check(lib.lspiv_memcpy_h2d(d_cam, _lib.ptr(a), a.nbytes))
        src = d_cam
        if self.normalize_samples:
            src = self._norm.ensure(T * n_cam)
            _lib.check(lib.lspiv_normalize_dev(d_cam, T, self.cam_shape[0], self.cam_shape[1], self.normalize_samples, src, None))
        src_dtype = np.uint8
        if self.edge_detect:
            d_edge = self._edge.ensure(T * n_cam * 4)
            # the recipe's minmax rides in the filter's store (lspiv_edge_detect_clip_dev: the bits of edge_detect + minmax, one pass)
            lo, hi = self.minmax if self.minmax else (float("-inf"), float("inf"))
            _lib.check(lib.lspiv_edge_detect_clip_dev(src, 0, T, self.cam_shape[0], self.cam_shape[1], self.edge_detect[0],
                                                      self.edge_detect[1], lo, hi, d_edge, None))
            src, src_dtype = d_edge, np.float32
        osz = 1 if self.ortho_uint8 else 4
        d_ortho = self._ortho.ensure(T * n_ortho * osz)
        self.projection.project_frames_dev(src.value, src_dtype, T, d_ortho.value, keep_uint8=self.ortho_uint8)
        d_out = self._out.ensure(4 * n_vec * 4)
        _lib.check(lib.lspiv_piv_pairs_dev(d_ortho, 0 if self.ortho_uint8 else 1, T, self.ortho_shape[0], self.ortho_shape[1], self.window_size[0],
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

    def _chunk_bounds(self, n_pairs: int, n_chunks: int):
        """Pair indices where the time chunks start (+ n_pairs): multiples of the kernels' anchor length, so that the chunked
        PIV returns the bits of one call (include/lspiv.h, lspiv_chunk_alignment)."""
        align = max(1, int(window.chunk_alignment(self.window_size, self.ortho_shape, self.overlap)))
        per = -(-n_pairs // max(1, int(n_chunks)))          # ceil
        per = max(align, -(-per // align) * align)
        return list(range(0, n_pairs, per)) + [n_pairs]

    def _run_streamed(self, a: np.ndarray, bounds, packed: bool):
        lib = _lib.load()
        T = a.shape[0]
        Hc, Wc = self.cam_shape
        n_cam, n_ortho, n_win = Hc * Wc, self.ortho_shape[0] * self.ortho_shape[1], self.n_rows * self.n_cols
        n_vec = (T - 1) * n_win
        d_cam = self._cam.ensure(T * n_cam)
        d_norm = self._norm.ensure(T * n_cam) if self.normalize_samples else None
        d_edge = self._edge.ensure(T * n_cam * 4) if self.edge_detect else None
        d_ortho = self._ortho.ensure(T * n_ortho * (1 if self.ortho_uint8 else 4))
        d_out = self._out.ensure(4 * n_vec * 4)             # chunk k's (4, pairs_k, n_win) block at float offset 4 * p_k * n_win
        d_pk = self._packed.ensure(4 * n_vec * 2) if packed else None
        if self._comp is None:
            self._comp = C.c_void_p()
            _lib.check(lib.lspiv_stream_create(C.byref(self._comp)))
        comp = self._comp
        at = lambda base, off: C.c_void_p(base.value + off)

        if self.normalize_samples:
            # the sampled frames first, each to its place in the stack; their mean plane is all normalize needs of the rest
            iv = round(T / self.normalize_samples)
            for t in range(0, T, iv):
                _lib.check(lib.lspiv_memcpy_h2d(at(d_cam, t * n_cam), _lib.ptr(a[t]), n_cam))
            d_mean = self._mean.ensure(n_cam * 4)
            _lib.check(lib.lspiv_normalize_mean_dev(d_cam, T, Hc, Wc, self.normalize_samples, d_mean, comp))
            _lib.check(lib.lspiv_stream_synchronize(comp))  # the chunks below overwrite those frames (with the same bytes)

        res = np.empty((4, T - 1, self.n_rows, self.n_cols), dtype=np.int16 if packed else np.float32)
        events = []
        try:
            self._stream_chunks(lib, a, bounds, packed, res, events, comp, d_cam, d_norm, d_edge, d_ortho, d_out, d_pk,
                                d_mean if self.normalize_samples else None)
        finally:
            # also on a failure half way (e.g. ENOMEM in a later chunk): nothing of this run may still be queued on the
            # compute stream when the buffers are reused by the next run() or freed by close(), and no event leaks
            lib.lspiv_stream_synchronize(comp)
            for ev in events:
                lib.lspiv_event_destroy(ev)
        return res[0], res[1], res[2], res[3]

    def _stream_chunks(self, lib, a, bounds, packed, res, events, comp, d_cam, d_norm, d_edge, d_ortho, d_out, d_pk, d_mean):
        Hc, Wc = self.cam_shape
        n_cam, n_ortho, n_win = Hc * Wc, self.ortho_shape[0] * self.ortho_shape[1], self.n_rows * self.n_cols
        at = lambda base, off: C.c_void_p(base.value + off)

        def fetch(k):                                       # result block of chunk k -> its rows of `res`
            p0, p1 = bounds[k], bounds[k + 1]
            blk = np.empty((4, p1 - p0, self.n_rows, self.n_cols), dtype=res.dtype)
            _lib.check(lib.lspiv_stream_wait_event(None, events[k]))
            src = at(d_pk, 4 * p0 * n_win * 2) if packed else at(d_out, 4 * p0 * n_win * 4)
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(blk), src, blk.nbytes))
            res[:, p0:p1] = blk

        for k in range(len(bounds) - 1):
            p0, p1 = bounds[k], bounds[k + 1]
            f0, f1 = (0 if k == 0 else p0 + 1), p1 + 1      # the frames this chunk brings in; frame p0 came with chunk k-1
            n_new = f1 - f0
            _lib.check(lib.lspiv_memcpy_h2d(at(d_cam, f0 * n_cam), _lib.ptr(a[f0:f1]), n_new * n_cam))   # blocking; chunk k-1 computes meanwhile
            src, src_dtype, esz = at(d_cam, f0 * n_cam), np.uint8, 1
            if self.normalize_samples:
                dst = at(d_norm, f0 * n_cam)
                _lib.check(lib.lspiv_normalize_apply_dev(src, n_new, Hc, Wc, d_mean, dst, comp))
                src = dst
            if self.edge_detect:
                dst = at(d_edge, f0 * n_cam * 4)
                lo, hi = self.minmax if self.minmax else (float("-inf"), float("inf"))
                _lib.check(lib.lspiv_edge_detect_clip_dev(src, 0, n_new, Hc, Wc, self.edge_detect[0], self.edge_detect[1], lo, hi, dst, comp))
                src, src_dtype, esz = dst, np.float32, 4
            osz = 1 if self.ortho_uint8 else 4
            self.projection.project_frames_dev(src.value, src_dtype, n_new, d_ortho.value + f0 * n_ortho * osz, comp.value,
                                               keep_uint8=self.ortho_uint8)
            blk_out = at(d_out, 4 * p0 * n_win * 4)
            _lib.check(lib.lspiv_piv_pairs_dev_at(at(d_ortho, p0 * n_ortho * osz), 0 if self.ortho_uint8 else 1, p1 - p0 + 1, self.ortho_shape[0],
                                                  self.ortho_shape[1], self.window_size[0], self.window_size[1], self.overlap[0],
                                                  self.overlap[1], self.signal_threshold, p0, blk_out, None, comp))
            if packed:
                _lib.check(lib.lspiv_pack_int16_dev(blk_out, 4 * (p1 - p0) * n_win, 0.01, -9999, at(d_pk, 4 * p0 * n_win * 2), comp))
            ev = C.c_void_p()
            _lib.check(lib.lspiv_event_create(C.byref(ev)))
            _lib.check(lib.lspiv_event_record_on(ev, comp))
            events.append(ev)
            if k >= 1:
                fetch(k - 1)
        fetch(len(bounds) - 2)

    def close(self):
        self.projection.close()
        for b in (self._cam, self._norm, self._edge, self._ortho, self._out, self._packed, self._mean):
            b.free()
        if self._comp is not None:
            _lib.load().lspiv_stream_destroy(self._comp)
            self._comp = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
