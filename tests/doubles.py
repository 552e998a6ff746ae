"""Stand-ins for the GPU calls in the CPU tests of the host mirrors: the same signatures, the oracle behind them."""
import numpy as np


def oracle_piv_pairs(fr, ws, ov, thr=None, pair_offset=0, out=None, scale=None):
    """``pyorc_amd.piv.piv_pairs`` (signature as of round 5, with ``out`` and ``scale``) computed by the C oracle; the px -> m/s scaling
    the way the reference writes it (ffpiv.py:418-419)."""
    from oracle import c_oracle

    res = tuple(a.astype(np.float32) for a in c_oracle.piv_pairs(np.asarray(fr), ws, ov, thr))
    if scale is not None:
        res_x, res_y, dt = scale
        dt = np.asarray(dt, dtype=np.float64)
        res = ((res[0] * res_x / np.expand_dims(dt, (1, 2))).astype(np.float32), (res[1] * res_y / np.expand_dims(dt, (1, 2))).astype(np.float32),
               res[2], res[3])
    if out is None:
        return res
    for dst, src in zip(out, res):
        assert dst.dtype == np.float32 and dst.shape == src.shape and dst.flags.c_contiguous
        dst[...] = src
    return tuple(out)


class HostStack:
    """``pyorc_amd.device.DeviceFrames`` as far as ``pyorc_amd.resident.ResidentStack`` and the oracle double above use it, backed by
    numpy: ``empty``, ``device_dtype``, ``upload`` (float64 narrowed to float32: what ``lspiv_upload_frames`` does for 8-bit-like
    imagery), time slices as views, ``from_host``.  ``uploads`` records (first frame, number of frames) of every piece."""

    uploads = []

    def __init__(self, arr):
        self.arr = arr
        self.shape, self.dtype = arr.shape, arr.dtype

    @classmethod
    def empty(cls, shape, dtype=np.uint8):
        return cls(np.zeros(shape, dtype))

    @classmethod
    def from_host(cls, frames):
        return cls(np.ascontiguousarray(frames))

    @staticmethod
    def device_dtype(host_dtype):
        dt = np.dtype(host_dtype)
        return dt if dt in (np.dtype(np.uint8), np.dtype(np.float32)) else np.dtype(np.float32)

    def upload(self, f0, frames, signal_threshold=None):
        a = np.asarray(frames)
        assert a.shape[1:] == self.shape[1:] and self.device_dtype(a.dtype) == self.dtype
        self.arr[f0:f0 + len(a)] = a.astype(self.dtype)
        HostStack.uploads.append((int(f0), len(a)))
        return len(a)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        assert isinstance(key, slice)
        return HostStack(self.arr[key])

    def __array__(self, dtype=None, copy=None):
        return self.arr if dtype is None else self.arr.astype(dtype)


def use_host_stacks(monkeypatch):
    """Route ``pyorc_amd.resident`` to :class:`HostStack` (CPU tests of the lazy path: no HBM here)."""
    from pyorc_amd import resident

    HostStack.uploads = []
    monkeypatch.setattr(resident, "DeviceFrames", HostStack)
    return HostStack
