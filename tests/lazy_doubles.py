"""A LAZY double of the xarray + dask pair as far as pyorc's ``project`` -> ``get_piv`` flow touches it (xarray is not installed in the
build image; dask only under /opt/conda -- tests/test_real_dask.py runs the same host code against the real one there).

``LazyDataArray`` is to ``tests/fake_xarray.DataArray`` what a dask-backed ``xr.DataArray`` is to an in-memory one:

* the time axis is cut into BLOCKS (pyorc reads videos in blocks of 20 frames, pyorc/api/video.py:48,528); ``.chunks`` reports them;
* ``.load()`` / ``.values`` compute every block the selection overlaps -- WHOLE blocks, on a thread pool (dask's threaded scheduler),
  each through the chain of per-block functions that built the array -- and count the calls per (layer, block): the tests assert that
  no block is computed twice;
* ``.data`` is a dask-array-like handle: ``name`` (``"<function>-<token>"``, dask's naming), ``chunks`` and ``dask.dependencies`` (a
  ``HighLevelGraph``'s layer-name -> set of layer names), which is what ``pyorc_amd.plugin`` reads to recognise ``project_hip``'s node;
* ``fillna(v)`` adds the three element-wise layers xarray's ``duck_array_ops.fillna`` produces on a dask array
  (``where`` <- {``invert`` <- ``isnan`` <- x, x}); ``map_time(f, name)`` adds an arbitrary element-wise layer (anything ELSE between
  ``project`` and ``get_piv``);
* ``apply_ufunc`` below is ``xr.apply_ufunc(..., dask="parallelized")`` for one input and one output with core dims (y, x).
"""
import itertools
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from tests import fake_xarray

_token = itertools.count(1)


def _name(prefix):
    return f"{prefix}-{next(_token):032x}"


_POOL = None


def _shared_pool():
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix="fake-dask")
    return _POOL


class Graph:
    def __init__(self, dependencies):
        self.dependencies = dependencies        # layer name -> set of layer names (dask.highlevelgraph.HighLevelGraph.dependencies)
        self.layers = {k: None for k in dependencies}


class ArrayHandle:
    """What ``xr.DataArray.data`` is for a dask-backed array, as far as the host code looks: name, chunks, dask (the graph)."""

    def __init__(self, name, chunks, graph):
        self.name, self.chunks, self.dask = name, chunks, graph


class LazyDataArray:
    def __init__(self, source, blocks, tail_shape, dtype, dims=("time", "y", "x"), coords=None, attrs=None, layers=(), name=None,
                 deps=None, lo=0, hi=None, calls=None, seconds_per_block=0.0, pool=None):
        self._source = source                          # numpy (T, ...) array: what the first layer's blocks are cut from
        self._blocks = list(blocks)                    # [0, b1, ..., T]
        self._tail, self._dtype = tuple(tail_shape), np.dtype(dtype)
        self.dims, self.attrs = tuple(dims), dict(attrs or {})
        self.coords = {k: (v if isinstance(v, fake_xarray.DataArray) else fake_xarray.DataArray(v, dims=(k,))) for k, v in (coords or {}).items()}
        self._layers = tuple(layers)                   # [(layer name, function of a block)], applied in order to a source block
        self._name = name or _name("from-video")
        self._deps = dict(deps) if deps is not None else {self._name: set()}
        self._lo, self._hi = lo, self._blocks[-1] if hi is None else hi
        self.calls = calls if calls is not None else {}   # (layer name, block index) -> number of computations
        self._calls_lock = threading.Lock()
        self.seconds_per_block = seconds_per_block
        self._pool = pool

    # ---- what the host code reads ----------------------------------------------------------------------------------------------
    dtype = property(lambda self: self._dtype)
    shape = property(lambda self: (self._hi - self._lo,) + self._tail)
    ndim = property(lambda self: 1 + len(self._tail))

    def __len__(self):
        return self._hi - self._lo

    @property
    def chunks(self):
        inside = [b for b in self._blocks if self._lo < b < self._hi]
        edges = [self._lo] + inside + [self._hi]
        return (tuple(b - a for a, b in zip(edges, edges[1:])),) + tuple((n,) for n in self._tail)

    @property
    def data(self):
        return ArrayHandle(self._name, self.chunks, Graph(self._deps))

    def _like(self, **kw):
        args = dict(source=self._source, blocks=self._blocks, tail_shape=self._tail, dtype=self._dtype, dims=self.dims, coords=self.coords,
                    attrs=self.attrs, layers=self._layers, name=self._name, deps=self._deps, lo=self._lo, hi=self._hi, calls=self.calls,
                    seconds_per_block=self.seconds_per_block, pool=self._pool)
        args.update(kw)
        return LazyDataArray(**args)

    def __getitem__(self, key):
        if isinstance(key, str):
            c = self.coords[key]
            return c[self._lo:self._hi] if c.dims == (self.dims[0],) and len(c) == self._blocks[-1] else c
        if isinstance(key, slice):
            a, b, step = key.indices(len(self))
            assert step == 1
            out = self._like(lo=self._lo + a, hi=self._lo + max(a, b), name=_name("getitem"))
            out._deps = {**self._deps, out._name: {self._name}}
            out._sliced_from = self._name
            return out
        k = int(key) + (len(self) if key < 0 else 0)
        return fake_xarray.DataArray(np.zeros(self._tail, self._dtype), self.dims[1:])     # frames[0].shape is all the host code wants

    # ---- computing ---------------------------------------------------------------------------------------------------------------
    def _block(self, i):
        import time as _time

        a, b = self._blocks[i], self._blocks[i + 1]
        out = np.asarray(self._source[a:b])
        if self.seconds_per_block:
            _time.sleep(self.seconds_per_block)
        for lname, fn in self._layers:
            with self._calls_lock:
                self.calls[(lname, i)] = self.calls.get((lname, i), 0) + 1
            out = fn(out)
        return out

    def load(self):
        idx = [i for i in range(len(self._blocks) - 1) if self._blocks[i] < self._hi and self._blocks[i + 1] > self._lo]
        parts = list((self._pool or _shared_pool()).map(self._block, idx))      # dask's threaded scheduler: blocks on worker threads
        # one block: its array as it is (dask hands a single chunk over without a copy); several: concatenated into a fresh array
        whole = parts[0] if len(parts) == 1 else (np.concatenate(parts) if parts else np.zeros((0,) + self._tail, self._dtype))
        first = self._blocks[idx[0]] if idx else self._lo
        vals = whole[self._lo - first:self._hi - first]
        coords = {k: (v[self._lo:self._hi] if v.dims == (self.dims[0],) and len(v) == self._blocks[-1] else v) for k, v in self.coords.items()}
        return fake_xarray.DataArray(np.ascontiguousarray(vals), self.dims, coords, self.attrs)

    values = property(lambda self: self.load().values)

    # ---- graph-building operations -------------------------------------------------------------------------------------------------
    def map_time(self, fn, prefix, tail_shape=None, dtype=None):
        """An element-wise (per block) layer on top."""
        name = _name(prefix)
        return self._like(layers=self._layers + ((name, fn),), name=name, deps={**self._deps, name: {self._name}},
                          tail_shape=self._tail if tail_shape is None else tail_shape, dtype=self._dtype if dtype is None else dtype)

    def fillna(self, value):
        """xarray's ``duck_array_ops.fillna(data, other) = where(notnull(data), data, other)`` on a dask array: isnan, invert, where."""
        p = self._name
        n_isnan, n_inv, n_where = _name("isnan"), _name("invert"), _name("where")
        deps = {**self._deps, n_isnan: {p}, n_inv: {n_isnan}, n_where: {n_inv, p}}
        fn = lambda blk: np.where(~np.isnan(blk), blk, np.asarray(value, dtype=blk.dtype))    # noqa: E731
        return self._like(layers=self._layers + ((n_where, fn),), name=n_where, deps=deps)

    def rename(self, names):
        out = self._like(dims=tuple(names.get(d, d) for d in self.dims))
        out.coords = {names.get(k, k): v for k, v in self.coords.items()}
        return out

    def __setitem__(self, key, value):
        self.coords[key] = value if isinstance(value, fake_xarray.DataArray) else fake_xarray.DataArray(value, dims=(key,))

    def diff(self, dim):
        raise NotImplementedError


def from_frames(frames, block=20, coords=None, attrs=None, seconds_per_block=0.0):
    """A lazy ``(time, y, x)`` stack over an in-memory array, in blocks of ``block`` frames (a video opened by pyorc)."""
    T = len(frames)
    blocks = list(range(0, T, block)) + [T]
    coords = dict(coords or {})
    coords.setdefault("time", np.arange(T) / 30.0)
    return LazyDataArray(frames, blocks, frames.shape[1:], frames.dtype, coords=coords, attrs=attrs, seconds_per_block=seconds_per_block)


def apply_ufunc(func, da, kwargs=None, input_core_dims=None, output_core_dims=None, dask_gufunc_kwargs=None, output_dtypes=None,
                vectorize=False, exclude_dims=frozenset(), dask="forbidden", keep_attrs=False):
    """``xr.apply_ufunc(..., dask="parallelized")`` for a lazy input (one input, one output, core dims last): a per-block layer named
    after the function, like ``dask.array.apply_gufunc`` names its blockwise layer; an in-memory input goes to the eager double."""
    if not isinstance(da, LazyDataArray):
        return fake_xarray.apply_ufunc(func, da, kwargs, input_core_dims, output_core_dims, dask_gufunc_kwargs, output_dtypes, vectorize,
                                       exclude_dims, dask, keep_attrs)
    assert dask == "parallelized" and not vectorize and list(input_core_dims[0]) == list(da.dims[-2:])
    sizes = (dask_gufunc_kwargs or {})["output_sizes"]
    core_out = list(output_core_dims[0])
    kw = kwargs or {}
    out = da.map_time(lambda blk: func(blk, **kw), getattr(func, "__name__", "gufunc").lstrip("_"), tail_shape=tuple(sizes[d] for d in core_out),
                      dtype=output_dtypes[0])
    out.dims = da.dims[:-2] + tuple(core_out)
    out.coords = {k: v for k, v in da.coords.items() if v.dims and all(d in out.dims for d in v.dims)}
    if not keep_attrs:
        out.attrs = {}
    return out


# what ``import xarray as xr`` must offer to pyorc_amd.plugin.project_hip / pyorc_amd.velocimetry in the tests
DataArray = fake_xarray.DataArray
Dataset = fake_xarray.Dataset
concat = fake_xarray.concat


class CameraConfigDouble:
    """The two members of pyorc's CameraConfig that ``project_numpy`` / ``project_hip`` call (pyorc/project.py:196-199), answering with
    given index maps (``pyorc_amd.synth.projection_maps``)."""

    def __init__(self, maps):
        self.maps = maps

    def map_idx_img_ortho(self, x, y, z):
        return self.maps[0], self.maps[1]

    def map_mean_idx_img_ortho(self, x, y, z):
        return self.maps[2], self.maps[3], self.maps[4]


def frames_project(da, maps, dst_shape, proj_method, reducer="mean"):
    """What ``Frames.project`` does around its method lookup (pyorc/api/frames.py:240-265): axes, the projection method called as
    ``proj_method(self._obj, cc, x, y, z, reducer)``, then ``fillna(0.0)``."""
    y = np.flipud(np.linspace(0.005, 0.01 * (dst_shape[0] - 0.5), dst_shape[0]))
    x = np.linspace(0.005, 0.01 * (dst_shape[1] - 0.5), dst_shape[1])
    da_proj = proj_method(da, CameraConfigDouble(maps), x, y, 1.25, reducer)
    return da_proj.fillna(0.0)
