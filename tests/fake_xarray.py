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

    data = property(lambda self: self.values)       # xarray: the backing array (numpy when in memory, dask when lazy)
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

    def rename(self, names):
        """xarray.DataArray.rename({old: new}): dimension (and coordinate) names."""
        dims = tuple(names.get(d, d) for d in self.dims)
        coords = {names.get(k, k): (DataArray(v.values, tuple(names.get(d, d) for d in v.dims)) if isinstance(v, DataArray) else v)
                  for k, v in self.coords.items()}
        return DataArray(self.values, dims, coords, self.attrs)

    def __setitem__(self, key, value):
        """``da["y"] = y``: assign a coordinate along the dimension of that name."""
        assert isinstance(key, str)
        self.coords[key] = value if isinstance(value, DataArray) else DataArray(value, dims=(key,))

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


def apply_ufunc(func, da, kwargs=None, input_core_dims=None, output_core_dims=None, dask_gufunc_kwargs=None, output_dtypes=None,
                vectorize=False, exclude_dims=frozenset(), dask="forbidden", keep_attrs=False):
    """An EAGER stand-in for ``xarray.apply_ufunc`` with one input and one output (what pyorc's project_numpy and
    pyorc_amd.plugin.project_hip use): the input's core dimensions are moved to the end, ``func`` gets the whole array (``vectorize``
    False) or one core slice at a time (True), the result carries the loop dimensions followed by the output core dimensions.  In
    ``blocks`` > 1 pieces along the first loop dimension when ``da.attrs`` asks for it (``_blocks``): what dask's blocks look like to
    ``func``."""
    core_in, core_out = list(input_core_dims[0]), list(output_core_dims[0])
    loop = [d for d in da.dims if d not in core_in]
    order = [da.dims.index(d) for d in loop + core_in]
    a = np.transpose(da.values, order)
    kw = kwargs or {}
    if vectorize:
        lead = a.shape[:len(loop)]
        out = np.stack([func(a[i], **kw) for i in np.ndindex(*lead)]).reshape(lead + np.shape(func(a[(0,) * len(lead)], **kw)))
    else:
        n_blocks = int(da.attrs.get("_blocks", 1)) if loop else 1
        pieces = np.array_split(np.arange(a.shape[0]), n_blocks) if loop else [None]
        out = np.concatenate([func(a[p[0]:p[-1] + 1], **kw) for p in pieces if len(p)]) if loop else func(a, **kw)
    if output_dtypes:
        assert out.dtype == np.dtype(output_dtypes[0]), (out.dtype, output_dtypes)
    sizes = (dask_gufunc_kwargs or {}).get("output_sizes", {})
    for d, n in zip(core_out, out.shape[len(loop):]):
        assert sizes.get(d, n) == n
    coords = {k: v for k, v in da.coords.items() if isinstance(v, DataArray) and v.dims and all(d in loop for d in v.dims)}
    return DataArray(out, tuple(loop + core_out), coords, da.attrs if keep_attrs else None)
