"""Mirror of the ``ffpiv`` API that pyorc's PIV wrapper binds, executed on the MI355X.

``cross_corr`` and ``u_v_displacement`` keep the signatures observed at the reference call
sites (pyorc/velocimetry/ffpiv.py:222-231, 324, 450-459, 471) so they can be swapped in for
``from ffpiv import cross_corr, u_v_displacement`` (ffpiv.py:9).  ``piv_pairs`` is the fused
entry the ``hip`` engine actually uses: one kernel produces u, v, corr_max and s2n per
window without ever writing the (T-1, n_win, wy, wx) correlation volume to memory.

No CPU fallback: every function needs liblspiv_hip.so and a gfx950 device.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib, window
from .device import is_device

ENGINES = ("hip",)


def _sig(signal_threshold: Optional[float]) -> float:
    return -1.0 if signal_threshold is None else float(signal_threshold)


def _check_args(window_size, overlap, search_area_size, normalize, engine):
    if engine not in ENGINES:
        raise ValueError(f"Selected PIV engine {engine} does not exist.")
    if normalize:
        raise NotImplementedError("stack-level `normalize` is never used by pyorc (ffpiv.py:227,455)")
    sa = tuple(window_size) if search_area_size is None else tuple(search_area_size)
    if sa != tuple(window_size):
        raise NotImplementedError("search_area_size must equal window_size (pyorc/api/frames.py:168)")


def piv_pairs(imgs, window_size=(32, 32), overlap=(16, 16), signal_threshold: Optional[float] = None,
              return_planes: bool = False, pair_offset: int = 0, out=None, scale=None):
    """Fused PIV of every consecutive frame pair of ``imgs`` (T, H, W).

    Returns ``(u, v, corr_max, s2n[, planes])``: float32 arrays (T-1, n_rows, n_cols); u, v in
    pixels (u = column shift, v = row shift).  Replaces pyorc/velocimetry/ffpiv.py:446-474.
    ``pair_offset``: index of the chunk's first pair in the whole stack (``lspiv_piv_pairs_at``); chunks that start on
    multiples of ``window.chunk_alignment`` reproduce the whole-stack result bit for bit.
    ``out``: four C-contiguous float32 arrays (T-1, n_rows, n_cols) that receive u, v, corr_max, s2n (e.g. time slices of the
    arrays of a whole run: the caller's chunk loop then needs no concatenation); they are what is returned.
    ``scale = (res_x, res_y, dt)``: u, v come back in metres per second, ``(u * res_x / dt[:, None, None]).astype(float32)`` (ffpiv.py:
    418-419) computed on the device before the results cross PCIe -- float32 product, float64 division, one rounding: numpy's own
    arithmetic for PYTHON-FLOAT resolutions (the caller checks that, :func:`device_scaling_is_numpys`); ``dt``: (T-1,) float64 seconds.
    """
    lib = _lib.load()
    _lib.require_device()
    a = imgs if is_device(imgs) else _lib.as_frames(imgs)
    T, H, W = a.shape
    n_rows, n_cols = window.get_array_shape((H, W), window_size, overlap)
    if T < 2 or n_rows < 1 or n_cols < 1:
        raise ValueError(f"need >= 2 frames at least one window large, got {a.shape} for window {window_size}")
    P = T - 1
    if out is not None:
        out = list(out)
        if len(out) != 4 or any(not isinstance(o, np.ndarray) or o.dtype != np.float32 or o.shape != (P, n_rows, n_cols) or
                                not o.flags.c_contiguous or not o.flags.writeable for o in out):
            raise ValueError(f"out must be four writable C-contiguous float32 arrays of shape {(P, n_rows, n_cols)}")
    dt = None
    if scale is not None:
        if return_planes:
            raise ValueError("scale and return_planes exclude each other")
        dt = np.ascontiguousarray(scale[2], dtype=np.float64).reshape(-1)
        if dt.shape != (P,):
            raise ValueError(f"scale: dt must have one entry per frame pair ({P}), got shape {dt.shape}")
    if is_device(a):   # HBM-resident stack: no staging, one launch, only the result block crosses PCIe
        return _piv_pairs_device(a, window_size, overlap, signal_threshold, return_planes, pair_offset, n_rows, n_cols, out,
                                 None if scale is None else (float(scale[0]), float(scale[1]), dt))
    if out is None:
        out = [np.empty((P, n_rows, n_cols), dtype=np.float32) for _ in range(4)]
    planes = None
    if return_planes:
        planes = np.empty((P, n_rows * n_cols, window_size[0], window_size[1]), dtype=np.float32)
    if scale is not None:
        _lib.check(lib.lspiv_piv_velocity_at(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], T, H, W, window_size[0], window_size[1],
                                             overlap[0], overlap[1], _sig(signal_threshold), int(pair_offset), float(scale[0]), float(scale[1]),
                                             _lib.ptr(dt), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(out[3])))
        return tuple(out)
    _lib.check(lib.lspiv_piv_pairs_at(_lib.ptr(a), _lib.DTYPE_CODES[a.dtype], T, H, W, window_size[0], window_size[1],
                                      overlap[0], overlap[1], _sig(signal_threshold), int(pair_offset),
                                      _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(out[3]),
                                      _lib.ptr(planes) if planes is not None else None))
    return (*out, planes) if return_planes else tuple(out)


def device_scaling_is_numpys(res_x, res_y) -> bool:
    """Is ``(u * res / dt).astype(float32)`` with THESE resolutions what the device computes (float32 product)?  numpy multiplies a
    float32 array by a Python float in float32; by a numpy float64 scalar in float64 (numpy >= 2) -- then the host keeps numpy's own
    arithmetic.  (``np.float64`` is a subclass of ``float``: the test is on the exact type.)"""
    return type(res_x) in (float, int, np.float32) and type(res_y) in (float, int, np.float32)


def _piv_pairs_device(a, window_size, overlap, signal_threshold, return_planes, pair_offset, n_rows, n_cols, out=None, scale=None):
    from .device import DeviceFrames

    lib = _lib.load()
    T, H, W = a.shape
    P, n_win = T - 1, n_rows * n_cols
    d_out = DeviceFrames.empty((4, P, n_win), np.float32)
    d_planes = DeviceFrames.empty((P * n_win, window_size[0], window_size[1]), np.float32) if return_planes else None
    _lib.check(lib.lspiv_piv_pairs_dev_at(a.c_ptr, a.dtype_code, T, H, W, window_size[0], window_size[1], overlap[0], overlap[1],
                                          _sig(signal_threshold), int(pair_offset), d_out.c_ptr,
                                          d_planes.c_ptr if d_planes is not None else None, None))
    if scale is not None:  # px / frame -> m / s in place on the [u | v] blocks (lspiv_scale_velocity_dev: the same kernel as the host entry point's)
        _lib.check(lib.lspiv_scale_velocity_dev(d_out.c_ptr, P, n_win, scale[0], scale[1], _lib.ptr(scale[2]), None))
    if out is not None:    # straight into the caller's arrays: [u | v | corr | s2n] are four consecutive (P, n_win) blocks
        for k in range(4):
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(out[k]), C.c_void_p(d_out.ptr + k * P * n_win * 4), out[k].nbytes))
        out = tuple(out)
    else:
        res = d_out.to_host().reshape(4, P, n_rows, n_cols)     # lspiv_memcpy_d2h runs on, and waits for, the library's stream
        out = tuple(np.ascontiguousarray(res[k]) for k in range(4))
    if return_planes:
        return (*out, d_planes.to_host().reshape(P, n_win, window_size[0], window_size[1]))
    return out


def cross_corr(imgs, window_size=(64, 64), overlap=(32, 32), search_area_size=None, normalize=False,
               engine="hip", signal_threshold=None, verbose=False):
    """``ffpiv.cross_corr`` drop-in: returns ``(x, y, corr)``, corr (T-1, n_win, wy, wx) float32.

    Planes of window pairs below ``signal_threshold`` are NaN (pyorc/velocimetry/ffpiv.py:93-97).
    """
    _check_args(window_size, overlap, search_area_size, normalize, engine)
    a = _lib.as_frames(imgs)
    x, y = window.get_rect_coordinates(a.shape[-2:], window_size, overlap)
    *_, planes = piv_pairs(a, window_size, overlap, signal_threshold, return_planes=True)
    return x, y, planes


def u_v_displacement(corr, n_rows: int, n_cols: int, engine: str = "hip") -> Tuple[np.ndarray, np.ndarray]:
    """``ffpiv.u_v_displacement`` drop-in: corr (P, n_win, wy, wx) -> u, v (P, n_rows, n_cols) in pixels."""
    if engine not in ENGINES:
        raise ValueError(f"Selected PIV engine {engine} does not exist.")
    lib = _lib.load()
    _lib.require_device()
    c = np.ascontiguousarray(corr, dtype=np.float32)
    if c.ndim == 3:
        c = c[None]
    P, n_win, wy, wx = c.shape
    if n_win != n_rows * n_cols:
        raise ValueError(f"corr has {n_win} windows, expected {n_rows}*{n_cols}")
    u = np.empty((P, n_rows, n_cols), dtype=np.float32)
    v = np.empty((P, n_rows, n_cols), dtype=np.float32)
    _lib.check(lib.lspiv_u_v_displacement(_lib.ptr(c), P, n_win, wy, wx, _lib.ptr(u), _lib.ptr(v)))
    return u, v


class Ensemble:
    """Device-resident ensemble-correlation accumulator (pyorc/velocimetry/ffpiv.py:182-376)."""

    def __init__(self, dim_size, window_size, overlap):
        lib = _lib.load()
        _lib.require_device()
        self._h = C.c_void_p()
        self.window_size = tuple(window_size)
        self.overlap = tuple(overlap)
        self.dim_size = tuple(dim_size)
        self.n_rows, self.n_cols = window.get_array_shape(dim_size, window_size, overlap)
        _lib.check(lib.lspiv_ensemble_begin(dim_size[0], dim_size[1], window_size[0], window_size[1],
                                            overlap[0], overlap[1], C.byref(self._h)))
        self._held = []   # DeviceFrames chunks the handle borrows until finish (float64 rescue of the final fit)
        self._retain_mode = None   # set_retain not called yet: accumulate() of a DeviceFrames chunk picks RETAIN_BORROW

    RETAIN_NONE, RETAIN_COPY, RETAIN_BORROW = 0, 1, 2

    def set_retain(self, mode: int) -> None:
        """How ``accumulate_dev`` keeps the chunks for the float64 rescue of the final fit (include/lspiv.h,
        ``lspiv_ensemble_set_retain``): 0 nothing (float32 fits), 1 the handle copies them, 2 it borrows the caller's pointers."""
        _lib.check(_lib.load().lspiv_ensemble_set_retain(self._h, int(mode)))
        self._retain_mode = int(mode)
        # what was borrowed so far STAYS pinned by this object until close(): lspiv_ensemble_set_retain only stores the mode, the handle
        # keeps the earlier borrowed pointers (in COPY mode for good, in NONE mode until the next accumulate), and a finish without
        # another accumulate reads them -- dropping them here let the caller free HBM the final fit's float64 rescue still read

    def stats(self) -> dict:
        """Counters of the last ``finish``: windows flagged / re-evaluated in float64 / left with their float32 fit, chunks and
        bytes kept, and whether every chunk could be kept."""
        st = (C.c_int64 * 6)()
        _lib.check(_lib.load().lspiv_ensemble_stats(self._h, st))
        return {"flagged": int(st[0]), "rescued": int(st[1]), "float32_kept": int(st[2]), "chunks_kept": int(st[3]),
                "bytes_kept": int(st[4]), "retain_complete": bool(st[5])}

    def accumulate(self, imgs, corr_min: float, s2n_min: float, signal_threshold: Optional[float] = None, out=None):
        """Add one frame chunk; returns masked per-pair (corr_max, s2n), each (T-1, n_win) float32.  ``out``: two C-contiguous float32
        arrays of that shape to receive them (time slices of a whole run's arrays: no concatenation afterwards)."""
        a = imgs if is_device(imgs) else _lib.as_frames(imgs)
        if tuple(a.shape[1:]) != self.dim_size:
            raise ValueError(f"chunk shape {a.shape[1:]} != ensemble shape {self.dim_size}")
        P = a.shape[0] - 1
        n_win = self.n_rows * self.n_cols
        if out is not None:
            out = tuple(out)
            if len(out) != 2 or any(not isinstance(o, np.ndarray) or o.dtype != np.float32 or o.shape != (P, n_win) or
                                    not o.flags.c_contiguous or not o.flags.writeable for o in out):
                raise ValueError(f"out must be two writable C-contiguous float32 arrays of shape {(P, n_win)}")
        if is_device(a):
            from .device import DeviceFrames

            d = DeviceFrames.empty((2, P, n_win), np.float32)
            # an HBM-resident stack stays alive as long as this object holds it: unless the caller chose a mode (set_retain: it is respected),
            # the handle borrows the pointer (no copy) and the final fit can go back to the frames.  The borrowed bytes count
            # against the handle's budget (LSPIV_ENSEMBLE_RETAIN_BYTES, a quarter of the HBM by default): beyond it the handle
            # gives the rescue up (stats()["retain_complete"] False, float32 fits) and this object lets go of every chunk --
            # a caller streaming more chunks through one ensemble than the budget holds is not pinned into an out-of-memory
            if self._retain_mode is None:      # no choice made: borrow
                _lib.check(_lib.load().lspiv_ensemble_set_retain(self._h, self.RETAIN_BORROW))
                self._retain_mode = self.RETAIN_BORROW
            if self._retain_mode == self.RETAIN_BORROW:
                self._held.append(a)
            self.accumulate_dev(a.ptr, a.dtype, a.shape[0], corr_min, s2n_min, d.ptr, signal_threshold)
            if self._held and not self.stats()["retain_complete"]:
                self._held = []
            if out is not None:
                for k in range(2):
                    _lib.check(_lib.load().lspiv_memcpy_d2h(_lib.ptr(out[k]), C.c_void_p(d.ptr + k * P * n_win * 4), out[k].nbytes))
                return out
            res = d.to_host()
            return np.ascontiguousarray(res[0]), np.ascontiguousarray(res[1])
        cm, sn = out if out is not None else (np.empty((P, n_win), dtype=np.float32), np.empty((P, n_win), dtype=np.float32))
        _lib.check(_lib.load().lspiv_ensemble_accumulate(self._h, _lib.ptr(a), _lib.DTYPE_CODES[a.dtype], a.shape[0],
                                                         float(corr_min), float(s2n_min), _sig(signal_threshold),
                                                         _lib.ptr(cm), _lib.ptr(sn)))
        return cm, sn

    def accumulate_dev(self, d_frames: int, dtype, n_frames: int, corr_min: float, s2n_min: float,
                       d_corr_s2n: int, signal_threshold: Optional[float] = None, stream: Optional[int] = None) -> None:
        """Same on a chunk that already sits in HBM; d_corr_s2n receives [corr_max | s2n], 2*(T-1)*n_win float32."""
        _lib.check(_lib.load().lspiv_ensemble_accumulate_dev(
            self._h, C.c_void_p(d_frames), _lib.DTYPE_CODES[np.dtype(dtype)], n_frames, float(corr_min), float(s2n_min),
            _sig(signal_threshold), C.c_void_p(d_corr_s2n), C.c_void_p(stream) if stream else None))

    def finish(self, count_min: float, n_frames: float, return_mean: bool = False):
        """Count filter + mean plane + sub-pixel peak: u, v (1, n_rows, n_cols) px, corr_count (n_win,)."""
        n_win = self.n_rows * self.n_cols
        u = np.empty((1, self.n_rows, self.n_cols), dtype=np.float32)
        v = np.empty((1, self.n_rows, self.n_cols), dtype=np.float32)
        cnt = np.empty(n_win, dtype=np.float32)
        mean = np.empty((1, n_win) + self.window_size, dtype=np.float32) if return_mean else None
        _lib.check(_lib.load().lspiv_ensemble_finish(self._h, float(count_min), float(n_frames), _lib.ptr(u),
                                                     _lib.ptr(v), _lib.ptr(cnt),
                                                     _lib.ptr(mean) if mean is not None else None))
        return (u, v, cnt, mean) if return_mean else (u, v, cnt)

    # ---- the finish in three stages, for a sum spread over several handles (include/lspiv.h; pyorc_amd.shard.sharded_ensemble) ----
    PARTIAL_DOUBLES = 20

    def flag(self, count_min: float, n_frames: float) -> int:
        """Mean planes + float32 fits + the windows whose fit needs float64: how many (the same on every rank holding the same state)."""
        n = C.c_int64(0)
        _lib.check(_lib.load().lspiv_ensemble_flag(self._h, float(count_min), float(n_frames), C.byref(n)))
        self._n_rec = int(n.value)
        return self._n_rec

    def flag_digest(self) -> int:
        """64-bit digest of the last ``flag``'s sorted records (window, candidates): equal on ranks that flagged the same list."""
        d = C.c_uint64(0)
        _lib.check(_lib.load().lspiv_ensemble_flag_digest(self._h, C.byref(d)))
        return int(d.value)

    def partials(self):
        """(partials (n_records, 20) float64 over THIS handle's retained chunks, complete: bool)."""
        part = np.zeros((getattr(self, "_n_rec", 0), self.PARTIAL_DOUBLES), dtype=np.float64)
        ok = C.c_int(0)
        _lib.check(_lib.load().lspiv_ensemble_partials(self._h, _lib.ptr(part) if part.size else None, C.byref(ok)))
        return part, bool(ok.value)

    def finish_partials(self, partials, return_mean: bool = False):
        """``finish`` with the flagged windows fitted from ``partials`` summed over all handles that share the state."""
        n_win = self.n_rows * self.n_cols
        part = np.ascontiguousarray(partials, dtype=np.float64)
        if part.shape != (getattr(self, "_n_rec", 0), self.PARTIAL_DOUBLES):
            raise ValueError(f"partials shape {part.shape} != ({getattr(self, '_n_rec', 0)}, {self.PARTIAL_DOUBLES})")
        u = np.empty((1, self.n_rows, self.n_cols), dtype=np.float32)
        v = np.empty((1, self.n_rows, self.n_cols), dtype=np.float32)
        cnt = np.empty(n_win, dtype=np.float32)
        mean = np.empty((1, n_win) + self.window_size, dtype=np.float32) if return_mean else None
        _lib.check(_lib.load().lspiv_ensemble_finish_partials(self._h, _lib.ptr(part) if part.size else None, _lib.ptr(u), _lib.ptr(v),
                                                              _lib.ptr(cnt), _lib.ptr(mean) if mean is not None else None))
        return (u, v, cnt, mean) if return_mean else (u, v, cnt)

    def export_state(self):
        """(corr_sum (n_win, wy, wx) float32 in fft-shifted layout, corr_count (n_win,) float32) from HBM."""
        n_win = self.n_rows * self.n_cols
        s = np.empty((n_win,) + self.window_size, dtype=np.float32)
        k = np.empty(n_win, dtype=np.float32)
        _lib.check(_lib.load().lspiv_ensemble_export(self._h, _lib.ptr(s), _lib.ptr(k)))
        return s, k

    def import_state(self, corr_sum, corr_count, add: bool = False):
        """Replace (or add to) the running state, e.g. with the all-reduced sums of every rank's time block."""
        s = np.ascontiguousarray(corr_sum, dtype=np.float32)
        k = np.ascontiguousarray(corr_count, dtype=np.float32)
        n_win = self.n_rows * self.n_cols
        if s.size != n_win * self.window_size[0] * self.window_size[1] or k.size != n_win:
            raise ValueError("state shape does not match this ensemble")
        _lib.check(_lib.load().lspiv_ensemble_import(self._h, _lib.ptr(s), _lib.ptr(k), int(bool(add))))

    def close(self):
        if self._h:
            _lib.load().lspiv_ensemble_destroy(self._h)
            self._h = C.c_void_p()
            self._held = []   # borrowed stacks are released with the handle, not before (finish may be called again)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
