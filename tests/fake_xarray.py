"""A test double of the few xarray features the host mirrors touch (pyorc_amd/velocimetry.py, frames.py, mask.py).

xarray is not installed in the build image, so the ``xr.DataArray`` / ``xr.Dataset`` branches of the mirrors would
otherwise never run before they meet a real pyorc installation.  This is NOT xarray: just labelled arrays with
``values / dims / coords / attrs``, positional slicing along the first axis, ``diff``, ``load``, ``concat`` and a
Dataset that is a dict of them.  tests/test_host.py installs it as ``sys.modules["xarray"]`` for one test."""
import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, attrs=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
        self.coords = {k: (v if isinstance(v, DataArray) else DataArray(v, dims=(k,))) for k, v in (coords or {}).items()}
        self.attrs = dict(attrs or {})
        self.loaded = 0

    dtype = property(lambda self: self.values.dtype)
    shape = property(lambda self: self.values.shape)
    ndim = property(lambda self: self.values.ndim)

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        sub = self.values[key]
        if isinstance(key, slice):      # slicing along the first (time) axis keeps the labels
            lead = self.dims[0]
            coords = {k: (v[key] if v.dims == (lead,) else v) for k, v in self.coords.items()}
            return DataArray(sub, self.dims, coords, self.attrs)
        return DataArray(sub, self.dims[1:], {k: v for k, v in self.coords.items() if v.dims and v.dims[0] in self.dims[1:]})

    def load(self):
        self.loaded += 1
        return self

    def copy(self, data=None):
        """xarray.DataArray.copy(data=...): same dims / coords / attrs / encoding, new values."""
        out = DataArray(self.values.copy() if data is None else data, self.dims, self.coords, self.attrs)
        out.encoding = dict(getattr(self, "encoding", {}))
        return out

    def diff(self, dim):
        assert self.dims == (dim,)
        return DataArray(np.diff(self.values), self.dims, {dim: self.values[1:]})


class Dataset(dict):
    def __init__(self, data_vars=None, coords=None, attrs=None):
        super().__init__()
        self.coords = {k: (v if isinstance(v, DataArray) else DataArray(v, dims=(k,))) for k, v in (coords or {}).items()}
        self.attrs = dict(attrs or {})
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, tuple):
            dims, data = value
            value = DataArray(data, dims, {d: self.coords[d] for d in dims if d in self.coords})
        super().__setitem__(key, value)

    def __getitem__(self, key):
        return super().__getitem__(key) if key in self.keys() else self.coords[key]

    data_vars = property(lambda self: dict(self))


def concat(objs, dim):
    return DataArray(np.concatenate([np.asarray(o.values) for o in objs]), (dim,))
