"""HBM-resident frame stacks: what lets a recipe ``normalize -> project -> get_piv`` stay on the GPU.

In pyorc every stage hands a full host stack to the next one (``Frames.normalize`` uint8 -> ``Frames.project`` float64,
x8 the camera bytes -> ``Frames.get_piv``; pyorc/api/frames.py:279-306, 199-277, 114-197), and ``get_ffpiv`` loads every
time chunk into host memory (pyorc/velocimetry/ffpiv.py:13-21).  Host-fed, the MI355X engine is PCIe-bound at a few
thousand pairs/s (float64: ~4 k) against ~150 k pairs/s from HBM.  ``DeviceFrames`` is a (T, H, W) stack that lives
in HBM; the mirrors in ``pyorc_amd.filters``, ``pyorc_amd.project.Projection`` and ``pyorc_amd.frames.get_piv`` /
``velocimetry.get_ffpiv`` accept it wherever they accept a numpy stack, call the ``*_dev`` entry points of the C ABI
and hand a ``DeviceFrames`` (or, for ``get_piv``, the usual result) back:

    f = DeviceFrames.from_host(camera_frames_uint8)          # the only host -> device copy
    f = filters.normalize(f, 15)                             # uint8, HBM
    f = projection.project_frames(f)                         # float32 ortho frames, HBM
    ds = frames.get_piv(f, 32, time=t, resolution=0.01)      # results on the host

Slicing along time (``f[a:b]``) is a view (pointer arithmetic), which is all the chunk loop of ``get_ffpiv`` needs.
"""

from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib


class _Pool:
    """Caching allocator behind DeviceFrames: hipMalloc / hipFree of a 0.4 - 1.7 GB stack cost milliseconds each, and a
    recipe allocates one stack per stage per chunk.  Freed blocks are kept (bucketed by size, rounded up to 1/8 of a power
    of two) and handed out again; at most ``limit`` bytes stay cached (``LSPIV_POOL_BYTES``, default 16 GiB of the 288 GB;
    0 disables the cache), ``release()`` gives everything back."""

    def __init__(self):
        import os

        import threading

        self.limit = int(os.environ.get("LSPIV_POOL_BYTES", 16 << 30))
        self.free = {}      # bucket size -> [raw pointer values]
        self.cached = 0
        self._lock = threading.RLock()   # stacks are made and dropped on several threads (the chunk executor's loaders stage their pieces)

    @staticmethod
    def bucket(nbytes: int) -> int:
        n = max(int(nbytes), 256)
        step = 1 << max(n.bit_length() - 4, 8)      # 1/8 .. 1/16 of the size: <= 12.5 % of slack
        return (n + step - 1) // step * step

    def take(self, nbytes: int):
        b = self.bucket(nbytes)
        with self._lock:
            lst = self.free.get(b)
            if lst:
                self.cached -= b
                return C.c_void_p(lst.pop()), b
        p = C.c_void_p()
        rc = _lib.load().lspiv_dev_malloc(C.byref(p), b)
        if rc == _lib.LSPIV_ENOMEM and self.cached:   # give the cache back and retry once
            self.release()
            rc = _lib.load().lspiv_dev_malloc(C.byref(p), b)
        _lib.check(rc)
        return p, b

    def give(self, ptr: C.c_void_p, b: int):
        with self._lock:
            if self.cached + b <= self.limit:
                self.free.setdefault(b, []).append(ptr.value)
                self.cached += b
                return
        _lib.load().lspiv_dev_free(ptr)

    def release(self):
        lib = _lib.load()
        with self._lock:
            free, self.free, self.cached = self.free, {}, 0
        for lst in free.values():
            for v in lst:
                lib.lspiv_dev_free(C.c_void_p(v))


_pool = _Pool()


def release_pool() -> None:
    """Return every cached HBM block of the DeviceFrames allocator to the driver."""
    _lib.check(_lib.load().lspiv_synchronize())
    _pool.release()


class _Allocation:
    """One HBM block, handed back to the pool when the last stack / view that uses it goes away.  Work on the library's
    stream is stream-ordered, so a block may be reused by the next stage while the previous kernels still run."""

    def __init__(self, nbytes: int):
        _lib.require_device()
        self.ptr, self._bucket = _pool.take(nbytes)
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                _pool.give(self.ptr, self._bucket)
                self.ptr = C.c_void_p()
        except Exception:
            pass


class DeviceFrames:
    """(T, H, W) frame stack in HBM: ``shape``, ``dtype``, ``len()``, time slicing (views), ``to_host()``."""

    def __init__(self, alloc: _Allocation, offset: int, shape: Tuple[int, int, int], dtype):
        self._alloc, self._offset = alloc, int(offset)
        self.shape = (int(shape[0]), int(shape[1]), int(shape[2]))
        self.dtype = np.dtype(dtype)
        if self.dtype not in _lib.DTYPE_CODES:
            raise TypeError(f"device stacks are uint8 / float32 / float64, got {self.dtype}")

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype=np.uint8) -> "DeviceFrames":
        dt = np.dtype(dtype)
        return cls(_Allocation(int(np.prod(shape)) * dt.itemsize), 0, shape, dt)

    @classmethod
    def from_host(cls, frames) -> "DeviceFrames":
        a = _lib.as_frames(frames)
        d = cls.empty(a.shape, a.dtype)
        _lib.check(_lib.load().lspiv_memcpy_h2d(d.c_ptr, _lib.ptr(a), a.nbytes))
        return d

    @staticmethod
    def device_dtype(host_dtype) -> np.dtype:
        """The sample type a host stack of ``host_dtype`` has once it is in HBM: uint8 and float32 as they are, everything else
        float32 (float64 is narrowed while it is staged, like the PIV host entry points do: the kernels compute in float32)."""
        dt = np.dtype(host_dtype)
        return dt if dt in (np.dtype(np.uint8), np.dtype(np.float32)) else np.dtype(np.float32)

    def upload(self, f0: int, frames, signal_threshold=None) -> int:
        """Host frames ``(n, H, W)`` into frames ``[f0, f0 + n)`` of this stack, the way ``lspiv_piv_pairs`` brings a host stack in
        (``lspiv_upload_frames``: pinned ring + staging threads, float64 narrowed to float32 with the DC-offset guard that
        ``signal_threshold`` -- the one of the PIV call that will read the frames -- switches exactly as there).  Returns ``n``."""
        a = _lib.as_frames(frames)
        n = int(a.shape[0])
        if a.shape[1:] != self.shape[1:]:
            raise ValueError(f"frames are {a.shape[1:]}, the stack holds {self.shape[1:]}")
        if not 0 <= f0 <= self.shape[0] - n:
            raise IndexError(f"frames [{f0}, {f0 + n}) do not fit a stack of {self.shape[0]}")
        if self.device_dtype(a.dtype) != self.dtype:
            raise TypeError(f"{a.dtype} frames arrive as {self.device_dtype(a.dtype)} in HBM, the stack is {self.dtype}")
        if n:
            frame_bytes = self.shape[1] * self.shape[2] * self.dtype.itemsize
            _lib.check(_lib.load().lspiv_upload_frames(C.c_void_p(self.ptr + int(f0) * frame_bytes), _lib.ptr(a), _lib.DTYPE_CODES[a.dtype], n,
                                                       self.shape[1], self.shape[2], -1.0 if signal_threshold is None else float(signal_threshold)))
        return n

    # ---- array-like surface ------------------------------------------------------------------
    @property
    def ptr(self) -> int:
        return self._alloc.ptr.value + self._offset

    @property
    def c_ptr(self) -> C.c_void_p:
        return C.c_void_p(self.ptr)

    @property
    def ndim(self) -> int:
        return 3

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape)) * self.dtype.itemsize

    @property
    def dtype_code(self) -> int:
        return _lib.DTYPE_CODES[self.dtype]

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, key):
        """Time slices are views; an integer index gives a one-frame view (used for ``frames[0].shape``)."""
        if isinstance(key, (int, np.integer)):
            k = int(key) + (self.shape[0] if key < 0 else 0)
            if not 0 <= k < self.shape[0]:
                raise IndexError(key)
            return _FrameView(self.shape[1:], self.dtype)
        if not isinstance(key, slice):
            raise TypeError("DeviceFrames supports integer indices and contiguous time slices")
        a, b, step = key.indices(self.shape[0])
        if step != 1:
            raise ValueError("DeviceFrames slices must be contiguous (step 1)")
        b = max(a, b)
        frame_bytes = self.shape[1] * self.shape[2] * self.dtype.itemsize
        return DeviceFrames(self._alloc, self._offset + a * frame_bytes, (b - a,) + self.shape[1:], self.dtype)

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            _lib.check(_lib.load().lspiv_memcpy_d2h(_lib.ptr(out), self.c_ptr, out.nbytes))
        return out

    def __repr__(self) -> str:
        return f"DeviceFrames(shape={self.shape}, dtype={self.dtype}, hbm=0x{self.ptr:x})"


class _FrameView:
    """What ``frames[0]`` needs to be for the accessor code: something with ``.shape``."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype


def is_device(obj) -> bool:
    return isinstance(obj, DeviceFrames)
