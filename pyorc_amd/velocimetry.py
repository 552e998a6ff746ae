"""Mirror of ``pyorc/velocimetry/ffpiv.py`` for ``engine="hip"``.

``get_ffpiv`` keeps the reference's signature, time chunking with a 1-frame halo
(pyorc/velocimetry/ffpiv.py:140), result layout (``s2n``, ``corr``, ``v_x``, ``v_y`` on
``(time, y, x)``, time = stamp of the 2nd frame of each pair, float32) and warnings/exceptions,
but each chunk is ONE fused GPU call instead of cross_corr + numpy reductions +
u_v_displacement over a materialised (T-1, n_win, wy, wx) volume.

Differences, all deliberate and documented in DESIGN.md:
  * a LAZY stack (anything with ``load()``: pyorc's dask-backed DataArray) is kept resident in HBM while it runs (``pyorc_amd.resident``,
    round 6): what ``.load()`` materialises at a time is planned against free HOST memory like in the reference (``available_memory() /
    memory_factor``, ffpiv.py:129; same warning text), divided by the number of loads the chunk executor keeps in flight, cut on dask's
    own block boundaries and WITHOUT the reference's halo frame (no block is decoded + projected twice); the kernels are launched on
    their anchors as the frames arrive.  When the stack is the direct product of ``project_hip`` (``pyorc_amd.plugin``), the CAMERA
    frames are loaded and projected on the device: ortho frames never exist on the host;
  * for a materialised stack (numpy, ``DeviceFrames``) the chunk size is planned against free HBM, not host RAM, and rounded down to a multiple of
    ``window.chunk_alignment`` (>= one multiple): chunks then start on the anchors of the time-walking kernels'
    segments, so the result is the same, bit for bit, whatever chunk size the planner or the user picked -- as in the
    reference, which computes every window independently;
  * quirk Q1 (user ``chunksize`` raises NameError in the reference, ffpiv.py:127-140) is fixed;
  * quirk Q2 (chunks are computed twice, ffpiv.py:402-408) is not reproduced;
  * quirk Q3 (ensemble ``n_frames`` = number of CHUNKS, ffpiv.py:373) IS reproduced, because it
    changes results (the ``count_min`` filter).

Works on ``xarray.DataArray`` frames (returns ``xarray.Dataset``) when xarray is importable, and
on plain ``(T, H, W)`` numpy arrays (returns ``PivResult``, a dict with the same variable names).
"""

from __future__ import annotations

import warnings
from typing import Literal, Optional, Tuple

import numpy as np

from . import executor, piv, resident, window
from .device import is_device

try:  # xarray is optional: the GPU box image does not ship it
    import xarray as xr
except ImportError:  # pragma: no cover - depends on the environment
    xr = None

CHUNK_SIZE_ERROR = (
    "Chunk size with selected nr of chunks ({chunks}) is 2 or less. If you manually "
    "selected `chunks={chunks}` then consider increasing chunk size to at least 2, and preferrably more. If memory "
    "is limited, consider closing memory intensive applications. If pyorc crashes, then this is due to "
    " insufficient memory."
)
CHUNK_SIZE_WARNING = (
    "Memory availability is poor ({avail_mem} GB). Chunk size is automatically set to {chunksize} to avoid "
    "memory issues. If pyorc crashes, then this is due to insufficient memory. Consider to manually set a lower "
    "chunk size e.g using `get_piv(engine={engine}, chunk=2)` or `get_piv(engine={engine}, chunk=3)` or close "
    "memory intensive applications."
)
MAX_WINDOWS_PER_LAUNCH = 2**31 - 1


class PivResult(dict):
    """Stand-in for ``xarray.Dataset`` when xarray is absent: data variables + ``coords`` + ``dims``."""

    dims = ("time", "y", "x")

    def __init__(self, data_vars, coords):
        super().__init__(data_vars)
        self.coords = coords

    def mean_time(self):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            return {k: np.nanmean(v, axis=0) for k, v in self.items()}


def _is_xr(obj) -> bool:
    return xr is not None and isinstance(obj, xr.DataArray)


def load_frame_chunk(da):
    """pyorc/velocimetry/ffpiv.py:13-21: materialise a chunk, retry without its last frame on TypeError."""
    if not hasattr(da, "load"):
        return da
    try:
        return da.load()
    except TypeError:
        return load_frame_chunk(da[:-1])


def _values(da):
    if is_device(da):
        return da   # HBM-resident chunk (a view): handed to the *_dev entry points as it is
    return da.values if hasattr(da, "values") else np.asarray(da)


def plan_chunks(n_frames: int, req_mem: float, avail_mem: float, chunksize: Optional[int], engine: str,
                n_win: int = 1):
    """Chunk planner of pyorc/velocimetry/ffpiv.py:127-142 -> (chunksize, [(start, stop), ...])."""
    if chunksize is None:
        chunks = int((req_mem // avail_mem) + 1)
        chunksize = int(np.ceil(n_frames / chunks))
        if chunksize <= 5:
            warnings.warn(
                CHUNK_SIZE_WARNING.format(avail_mem=avail_mem / 1e9, chunksize=chunksize, engine=engine), stacklevel=3
            )
            chunksize = 5  # hard override, try to manage with 5
            chunks = int(np.ceil(n_frames / chunksize))
    else:
        chunksize = int(chunksize)
        chunks = int(np.ceil(n_frames / max(chunksize, 1)))  # reference: NameError here (quirk Q1)
    if chunksize < 2:
        raise OverflowError(CHUNK_SIZE_ERROR.format(chunks=chunks))
    # one launch indexes windows with 32 bits
    max_cs = max(2, MAX_WINDOWS_PER_LAUNCH // max(n_win, 1))
    if chunksize > max_cs:
        chunksize = max_cs
        chunks = int(np.ceil(n_frames / chunksize))
    slices = [(max(c * chunksize - 1, 0), min((c + 1) * chunksize, n_frames)) for c in range(chunks)]
    # check if there are chunks that are too small in size, needs to be at least 2 frames per chunk
    return chunksize, [(a, b) for a, b in slices if b - a >= 2]


def aligned_slices(n_frames: int, chunksize: int, align: int, n_win: int = 1, fits=None):
    """Frame slices ``[(a, b), ...]`` of chunks whose first PAIR index ``a`` is a multiple of ``align``.

    ``chunksize`` is the number of FRAMES per chunk planned by :func:`plan_chunks` (a memory bound).  A chunk of C pairs
    reads C + 1 frames -- the halo frame sits at the end of a chunk instead of at its start (ffpiv.py:140), the union of
    pairs is the same --, so C is ``chunksize - 1`` rounded down to a multiple of ``align``.  When the plan is smaller
    than one anchor length (the forced ``chunksize = 5`` of the low-memory branch, or a small user value), one anchor
    length is still used IF ``fits(n_frames_of_chunk)`` says it fits the memory budget (chunk-invariant results are worth
    more than a plan that was conservative for the reference's 4-15x larger window stack); otherwise the plan is
    honoured with chunks of ``chunksize - 1`` pairs that start off the anchors: correct, and equal to an aligned run
    up to the last float32 bit of a chunk's first pairs.
    """
    n_pairs = n_frames - 1
    budget = max(1, int(chunksize) - 1)            # pairs per chunk the plan allows (C + 1 <= chunksize frames)
    if budget >= align:
        C = (budget // align) * align
    elif fits is None or fits(min(align, n_pairs) + 1):
        C = align
    else:
        C = budget
    max_pairs = max(1, MAX_WINDOWS_PER_LAUNCH // max(n_win, 1))
    if C > max_pairs:  # one launch indexes windows with 32 bits: shrink, aligned if possible
        C = max(1, (max_pairs // align) * align) if max_pairs >= align else max_pairs
    return [(p, min(p + C, n_pairs) + 1) for p in range(0, n_pairs, C)]


def _dataset(data_vars, time, y, x, like):
    if _is_xr(like):
        return xr.Dataset({k: (["time", "y", "x"], v) for k, v in data_vars.items()},
                          coords={"time": time, "y": y, "x": x})
    return PivResult(data_vars, {"time": np.asarray(time), "y": np.asarray(y), "x": np.asarray(x)})


def get_ffpiv(
    frames,
    y: np.ndarray,
    x: np.ndarray,
    dt: np.ndarray,
    window_size: Tuple[int, int],
    overlap: Tuple[int, int],
    search_area_size: Tuple[int, int],
    res_y: float,
    res_x: float,
    chunksize: Optional[int] = None,
    memory_factor: float = 4,
    engine: Literal["hip"] = "hip",
    ensemble_corr: bool = False,
    corr_min: float = 0.2,
    s2n_min: float = 3,
    count_min: float = 0.2,
    signal_threshold: Optional[float] = None,
    time: Optional[np.ndarray] = None,
    prefetch: Optional[int] = None,
):
    """Compute time-resolved (or ensemble) PIV on the MI355X; signature of pyorc's ``get_ffpiv`` (ffpiv.py:24-42).

    ``frames``: ``xr.DataArray (time, y, x)``, a ``(T, H, W)`` array, or a ``pyorc_amd.device.DeviceFrames`` stack that
    already lives in HBM (the output of ``pyorc_amd.filters`` / ``Projection.project_frames`` on device stacks: no
    chunk is staged through the host then, each chunk is one launch on a view of the stack); ``dt``: time step per pair (``T-1``,
    seconds; an ``xr.DataArray`` on ``time[1:]`` in pyorc); ``time``: frame time stamps when ``frames`` is a
    plain array (default ``arange(T)``).  Returns Dataset / PivResult with ``s2n, corr, v_x, v_y``.

    ``prefetch``: how many lazy chunks are materialised (``load_frame_chunk``: dask runs decode + projection + filters there)
    AHEAD of the chunk being launched, on a worker thread (``pyorc_amd.executor``; default ``LSPIV_PREFETCH_DEPTH`` or 1; 0 = the
    reference's serial load-then-compute loop, ffpiv.py:399-408).  Same chunks, same order, same results; stacks that are already
    materialised (numpy, ``DeviceFrames``) have nothing to prefetch.
    """
    if engine != "hip":
        raise ValueError(f"Selected PIV engine {engine} does not exist.")
    if tuple(search_area_size) != tuple(window_size):
        raise NotImplementedError("search_area_size must equal window_size (pyorc/api/frames.py:168)")
    n_frames = len(frames)
    dim_size = tuple(frames[0].shape)
    dtype = frames.dtype if np.dtype(frames.dtype) in (np.dtype(np.uint8), np.dtype(np.float32)) else np.float64
    n_rows, n_cols = len(y), len(x)
    # compute memory availability and size of problem (HBM instead of host RAM)
    user_chunksize = chunksize
    req_mem = window.required_memory(n_frames=n_frames, dim_size=dim_size, window_size=window_size,
                                     overlap=overlap, search_area_size=search_area_size, dtype=dtype)
    avail_mem = window.available_memory() / memory_factor
    chunksize, ref_slices = plan_chunks(n_frames, req_mem, avail_mem, chunksize, engine, n_win=n_rows * n_cols)
    def fits(n_chunk_frames: int) -> bool:   # does a chunk of that many frames respect the planner's memory budget?
        return window.required_memory(n_frames=n_chunk_frames, dim_size=dim_size, window_size=window_size, overlap=overlap,
                                      search_area_size=search_area_size, dtype=dtype) <= avail_mem

    slices = aligned_slices(n_frames, chunksize, window.chunk_alignment(window_size, dim_size, overlap), n_win=n_rows * n_cols, fits=fits)
    if time is None:
        time = frames["time"] if _is_xr(frames) else np.arange(n_frames)
    dt_arr = np.asarray(_values(dt), dtype=np.float64)
    if dt_arr.shape != (n_frames - 1,):
        raise ValueError(f"dt must have one entry per frame pair ({n_frames - 1}), got shape {dt_arr.shape}")
    if _is_lazy(frames) and n_frames >= 2 and not _stack_signal_mode(signal_threshold):
        # a lazy stack: resident in HBM, loads planned against HOST memory and cut where dask cuts, launches on the anchors
        plan = plan_lazy(frames, n_frames, dim_size, window_size, overlap, n_rows * n_cols, user_chunksize, memory_factor, engine, prefetch)
        return _get_ffpiv_lazy(frames, plan, y, x, dt_arr, time, res_y, res_x, n_cols, n_rows, window_size, overlap, ensemble_corr,
                               corr_min, s2n_min, count_min, signal_threshold, ref_slices)
    frames_chunks = [frames[a:b] for a, b in slices]
    depth = (executor.default_depth() if prefetch is None else int(prefetch)) if hasattr(frames, "load") else 0
    args = (frames_chunks, slices, y, x, dt_arr, time, res_y, res_x, n_cols, n_rows, window_size, overlap)
    if ensemble_corr:
        # quirk Q3 (ffpiv.py:373): the count filter is scaled with the number of CHUNKS -- of the reference's planner
        # (same formula, fed with HBM figures), not of the aligned chunks actually launched
        return _get_ffpiv_mean(*args, corr_min, s2n_min, count_min, signal_threshold, like=frames,
                               ref_slices=ref_slices, prefetch=depth)
    return _get_ffpiv_timestep(*args, signal_threshold, like=frames, prefetch=depth)


def _to_velocity(disp: np.ndarray, res, dt_chunk: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
    """``(disp * res / dt).astype(float32)`` of ffpiv.py:418-419 with the same arithmetic (the product in whatever type numpy
    gives it, the division in float64, one rounding to float32) but without the two float64 temporaries of the one-liner;
    ``out``: where the result goes (a time slice of the run's result array)."""
    prod = disp * res
    if out is None:
        out = prod if prod.dtype == np.float32 else np.empty(prod.shape, dtype=np.float32)
    np.divide(prod, dt_chunk, out=out, dtype=np.float64, casting="same_kind")
    return out


def _get_ffpiv_timestep(frames_chunks, slices, y, x, dt, time, res_y, res_x, n_cols, n_rows, window_size, overlap,
                        signal_threshold, like=None, prefetch=0):
    """Per-chunk loop of pyorc/velocimetry/ffpiv.py:379-443 (one fused GPU call per chunk); chunk n + 1 is loaded on a worker
    thread while chunk n is launched (``pyorc_amd.executor.ChunkPrefetcher``; ``prefetch = 0``: the reference's serial order)."""
    # the four result variables of the WHOLE run are allocated once and every chunk's launch writes its time slice of them (corr, s2n
    # straight from the library, v_x / v_y through the px -> m/s scaling): no per-chunk arrays to concatenate at the end (the
    # reference's xr.concat, ffpiv.py:442 -- 125 KB per pair, i.e. as many bytes again as a uint8 chunk's upload at 32 x 32)
    n_total = slices[-1][1] - 1 if slices else 0
    names = ("s2n", "corr", "v_x", "v_y")
    full = None            # name -> (n_total, n_rows, n_cols) float32
    px = None              # scratch for a chunk's u, v in pixels
    done = []              # (first pair, one past the last pair) of every chunk that delivered
    times = []
    on_device = piv.device_scaling_is_numpys(res_x, res_y)
    loader = executor.ChunkPrefetcher(frames_chunks, load_frame_chunk, depth=prefetch)
    try:
        for n, da in loader:
            a, b = slices[n]
            if len(da) >= 2:  # we need at least one image-pair to do PIV
                nb = a + len(da)  # load_frame_chunk may have dropped trailing frames
                p = nb - 1 - a
                vals = _values(da)
                grid = window.get_array_shape(tuple(vals.shape[1:]), window_size, overlap)
                if tuple(grid) != (n_rows, n_cols):
                    raise ValueError(f"grid {tuple(grid)} does not match coordinates ({n_rows}, {n_cols})")
                if full is None:
                    full = {k: np.empty((n_total, n_rows, n_cols), dtype=np.float32) for k in names}
                dt_chunk = dt[a:nb - 1]  # dt.sel(time=da.time[1:]), ffpiv.py:403-404
                if on_device:
                    # u and v to meter per second on the device, before they cross PCIe: the arithmetic of ffpiv.py:418-419 for
                    # python-float resolutions (float32 product, float64 division, float32 storage)
                    piv.piv_pairs(vals, window_size, overlap, signal_threshold, pair_offset=a, scale=(res_x, res_y, dt_chunk),
                                  out=(full["v_x"][a:a + p], full["v_y"][a:a + p], full["corr"][a:a + p], full["s2n"][a:a + p]))
                else:
                    if px is None or px[0].shape[0] < p:
                        px = [np.empty((p, n_rows, n_cols), dtype=np.float32) for _ in range(2)]
                    u, v = px[0][:p], px[1][:p]
                    piv.piv_pairs(vals, window_size, overlap, signal_threshold, pair_offset=a,
                                  out=(u, v, full["corr"][a:a + p], full["s2n"][a:a + p]))
                    # ... on the host with numpy's own arithmetic for any other kind of resolution (a numpy float64 scalar makes the product float64)
                    _to_velocity(u, res_x, dt_chunk[:, None, None], out=full["v_x"][a:a + p])
                    _to_velocity(v, res_y, dt_chunk[:, None, None], out=full["v_y"][a:a + p])
                done.append((a, a + p))
                times.append(time[a + 1:nb])
            # remove chunk safely from memory.  (The reference follows this with gc.collect() to get rid of its window stack and
            # correlation volume, ffpiv.py:437-440; neither exists here, and a collection costs ~1 ms per chunk: dropped.)
            frames_chunks[n] = None
            del da
    finally:
        # also when a launch raises in the loop body: the worker threads stop loading the next chunks NOW, not when the traceback is
        # collected, and the statistics of what ran are kept
        loader.close()
        executor.LAST_STATS.clear(); executor.LAST_STATS.update(loader.stats)
    if not done:
        raise ValueError("no chunk with at least one frame pair")
    if done[0][0] == 0 and done[-1][1] == n_total and all(x[1] == y[0] for x, y in zip(done, done[1:])):
        data = full            # every pair delivered, in order: the arrays are the result
    else:                      # a chunk lost trailing frames (load_frame_chunk's TypeError retry) or was skipped: close the gaps
        data = {k: np.concatenate([full[k][i:j] for i, j in done], axis=0) for k in names}
    if _is_xr(like):
        t = xr.concat(times, dim="time")
    else:
        t = np.concatenate([np.asarray(tt) for tt in times])
    return _dataset(data, t, y, x, like)


def _get_ffpiv_mean(frames_chunks, slices, y, x, dt, time, res_y, res_x, n_cols, n_rows, window_size, overlap,
                    corr_min, s2n_min, count_min, signal_threshold, like=None, ref_slices=None, prefetch=0):
    """Ensemble correlation of pyorc/velocimetry/ffpiv.py:182-376; corr_sum / corr_count stay in HBM; chunks loaded ahead of
    their launches like in ``_get_ffpiv_timestep`` (the reference's loop: ffpiv.py:348-370)."""
    dim_size = None
    ens = None
    # the masked per-pair corr_max / s2n of the whole run: allocated once, every chunk writes its time slice (Ensemble.accumulate(out=))
    n_total = slices[-1][1] - 1 if slices else 0
    n_win = n_rows * n_cols
    full = None
    done = []
    t_first = None
    loader = executor.ChunkPrefetcher(frames_chunks, load_frame_chunk, depth=prefetch)
    try:
        for n, da in loader:
            a, b = slices[n]
            if len(da) < 2:
                continue
            arr = _values(da)
            p = len(da) - 1
            if ens is None:
                dim_size = tuple(arr.shape[1:])
                ens = piv.Ensemble(dim_size, window_size, overlap)
                full = (np.empty((n_total, n_win), dtype=np.float32), np.empty((n_total, n_win), dtype=np.float32))
            ens.accumulate(arr, corr_min, s2n_min, signal_threshold, out=(full[0][a:a + p], full[1][a:a + p]))
            done.append((a, a + p))
            t_first = time[a + 1:a + 2]
            frames_chunks[n] = None
            del da
        if ens is None:
            raise ValueError("no chunk with at least one frame pair")
        # quirk Q3: `n_frames` is the number of CHUNKS, not pairs (ffpiv.py:373), and `time[0:1]` of the LAST chunk ends
        # up on the result (ffpiv.py:336) -- both taken from the reference's own chunk plan, not from the aligned
        # chunks that were launched, so neither depends on the segment anchoring
        n_frames = len(done)
        if ref_slices:
            n_frames = len(ref_slices)
            t_first = time[ref_slices[-1][0] + 1:ref_slices[-1][0] + 2]
        u, v, corr_count = ens.finish(count_min, n_frames)
    finally:
        loader.close()
        if ens is not None:
            ens.close()
    executor.LAST_STATS.clear(); executor.LAST_STATS.update(loader.stats)
    if done[0][0] == 0 and done[-1][1] == n_total and all(x[1] == y[0] for x, y in zip(done, done[1:])):
        corr_max_concat, s2n_concat = full
    else:   # a chunk lost trailing frames (load_frame_chunk's TypeError retry) or was skipped
        corr_max_concat = np.concatenate([full[0][i:j] for i, j in done], axis=0)
        s2n_concat = np.concatenate([full[1][i:j] for i, j in done], axis=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        # very low amounts of found valid correlations are entirely filtered out (ffpiv.py:280-286)
        corr_max_concat[:, corr_count < count_min * n_frames] = np.nan
        corr_max_mean = np.nanmean(corr_max_concat, axis=0).reshape(-1, n_rows, n_cols)
        s2n_mean = np.nanmean(s2n_concat, axis=0).reshape(-1, n_rows, n_cols)
    dt_av = dt.mean()
    data = {
        "s2n": s2n_mean,
        "corr": corr_max_mean,
        "v_x": (u * res_x / dt_av).astype(np.float32),
        "v_y": (v * res_y / dt_av).astype(np.float32),
    }
    return _dataset(data, t_first, y, x, like)


# ---------------------------------------------------------------------------------------------------------------------------------
# lazy stacks (round 6): resident in HBM, loads against the host budget on dask's block boundaries, launches on the anchors
# ---------------------------------------------------------------------------------------------------------------------------------
def _is_lazy(frames) -> bool:
    """Something that materialises on ``load()``: pyorc's dask-backed DataArray.  An ``xr.DataArray`` that is already in memory has a
    ``load()`` too; its ``.data`` (the backing array: looking at it computes nothing) is a numpy array, and it runs as one."""
    if is_device(frames) or not hasattr(frames, "load"):
        return False
    return not isinstance(getattr(frames, "data", None), np.ndarray)


def _stack_signal_mode(signal_threshold) -> bool:
    """The "stack" reading of ``signal_threshold`` scores a window position over all frames of a CHUNK (include/lspiv.h, option
    ``signal_mode`` = 1): its results depend on the chunking by definition, so such a run keeps the chunk loop it was defined on."""
    if signal_threshold is None:
        return False
    from . import _lib

    try:
        return _lib.get_option("signal_mode") == 1
    except Exception:
        return False


class LazyPlan(dict):
    """What :func:`plan_lazy` decided (a dict, for the tests and ``executor.LAST_STATS``): ``windows`` [(w0, w1)] frame ranges resident at
    a time, ``loads`` per window [(f0, f1)], ``load_frames``, ``host_budget`` / ``host_frame_bytes`` / ``peak_host_bytes``, ``depth`` (None =
    adaptive) and ``max_depth``, ``align``, ``blocks`` (dask's), ``source`` ("frames" | "camera": the pre-projection stack of project_hip)."""


def plan_lazy(frames, n_frames, dim_size, window_size, overlap, n_win, chunksize, memory_factor, engine, prefetch,
              host_available=None, hbm_available=None) -> LazyPlan:
    """How a lazy stack runs.  The reference's planner (ffpiv.py:119-139) with its own quantities where they still mean something:

    * what a load materialises is bounded by the HOST: ``available_memory() / memory_factor`` (ffpiv.py:129) divided by the loads the
      executor may keep in flight (``max_depth + 1``, or ``prefetch + 1`` when the caller fixed it), in frames of the stack that is
      actually loaded (the camera frames when the stack is ``project_hip``'s product); at 5 frames or fewer the reference's warning is
      raised and 5 it is (ffpiv.py:131-137); a user ``chunksize`` is honoured as the load size;
    * below that bound, loads are the overlap granule of ``pyorc_amd.resident.load_size`` -- at least ``MIN_LOADS`` of them when the
      stack has that many anchors -- in whole dask blocks;
    * what is resident at a time is bounded by HBM (``lspiv_available_bytes / memory_factor``): normally everything."""
    from . import plugin

    align = window.chunk_alignment(window_size, dim_size, overlap)
    handoff = plugin.hip_projection_source(frames)
    src = frames if handoff is None else handoff["source"]
    src_shape = tuple(int(v) for v in (src[0].shape if handoff is not None else dim_size))
    host_frame_bytes = int(np.prod(src_shape)) * np.dtype(src.dtype).itemsize
    blocks = resident.time_blocks(src)
    if prefetch is None and executor.adaptive_by_default():
        depth, deepest = None, executor.max_depth()
    else:
        depth = executor.default_depth() if prefetch is None else int(prefetch)
        deepest = depth
    host_avail = window.available_host_memory() if host_available is None else host_available
    host_budget = host_avail / memory_factor
    if chunksize is None:
        host_frames = int(host_budget // ((deepest + 1) * host_frame_bytes))
        if host_frames <= 5:
            warnings.warn(CHUNK_SIZE_WARNING.format(avail_mem=host_budget / 1e9, chunksize=host_frames, engine=engine), stacklevel=4)
            host_frames = 5  # hard override, try to manage with 5
        load_frames = resident.load_size(n_frames, align, host_frames, blocks)
    else:
        load_frames = int(chunksize)
        if load_frames < 2:
            raise OverflowError(CHUNK_SIZE_ERROR.format(chunks=int(np.ceil(n_frames / max(load_frames, 1)))))
    # HBM: the narrowed stack + the result block (+ one load of camera frames next to it when they are projected on the device)
    dev_itemsize = 4 if handoff is not None else resident.DeviceFrames.device_dtype(frames.dtype).itemsize
    hbm_avail = (window.available_memory() if hbm_available is None else hbm_available) / memory_factor
    per_frame = int(np.prod(dim_size)) * dev_itemsize + 16 * n_win
    scratch = load_frames * host_frame_bytes if handoff is not None else 0
    frames_per_window = max(align + 1, int((hbm_avail - scratch) // per_frame))
    frames_per_window = min(frames_per_window, max(2, MAX_WINDOWS_PER_LAUNCH // max(n_win, 1)))
    windows = resident.hbm_windows(n_frames, frames_per_window, align, blocks)
    loads = [resident.plan_loads(w1, load_frames, blocks, first=w0) for w0, w1 in windows]
    biggest = max((b - a for ls in loads for a, b in ls), default=0)
    return LazyPlan(windows=windows, loads=loads, load_frames=load_frames, align=align, blocks=blocks, depth=depth, max_depth=deepest,
                    host_budget=host_budget, host_frame_bytes=host_frame_bytes, peak_host_bytes=(deepest + 1) * biggest * host_frame_bytes,
                    source="frames" if handoff is None else "camera", handoff=handoff)


def _get_ffpiv_lazy(frames, plan, y, x, dt, time, res_y, res_x, n_cols, n_rows, window_size, overlap, ensemble_corr,
                    corr_min, s2n_min, count_min, signal_threshold, ref_slices):
    """The loops of pyorc/velocimetry/ffpiv.py:348-370 and :399-440 over a lazy stack: pieces are loaded ahead (``ChunkPrefetcher``),
    pushed into an HBM-resident stack (``ResidentStack``) and launched on the anchors; results land in the run's arrays."""
    from . import plugin

    n_total = len(frames) - 1
    n_win = n_rows * n_cols
    dim_size = tuple(int(v) for v in frames[0].shape)
    handoff = plan["handoff"]
    lazy = frames if handoff is None else handoff["source"]
    projection = None if handoff is None else plugin.projection_for(handoff)
    names = ("s2n", "corr", "v_x", "v_y")
    done = []
    ens = None
    on_device = piv.device_scaling_is_numpys(res_x, res_y)
    if ensemble_corr:
        ens = piv.Ensemble(dim_size, window_size, overlap)
        full = (np.empty((n_total, n_win), dtype=np.float32), np.empty((n_total, n_win), dtype=np.float32))

        def launch(view, p0, p1):
            ens.accumulate(view, corr_min, s2n_min, signal_threshold, out=(full[0][p0:p1], full[1][p0:p1]))
            done.append((p0, p1))
    else:
        full = {k: np.empty((n_total, n_rows, n_cols), dtype=np.float32) for k in names}
        px = []

        def launch(view, p0, p1):
            p = p1 - p0
            if on_device:
                piv.piv_pairs(view, window_size, overlap, signal_threshold, pair_offset=p0, scale=(res_x, res_y, dt[p0:p1]),
                              out=(full["v_x"][p0:p1], full["v_y"][p0:p1], full["corr"][p0:p1], full["s2n"][p0:p1]))
            else:
                if not px or px[0].shape[0] < p:
                    px[:] = [np.empty((p, n_rows, n_cols), dtype=np.float32) for _ in range(2)]
                u, v = px[0][:p], px[1][:p]
                piv.piv_pairs(view, window_size, overlap, signal_threshold, pair_offset=p0, out=(u, v, full["corr"][p0:p1], full["s2n"][p0:p1]))
                _to_velocity(u, res_x, dt[p0:p1, None, None], out=full["v_x"][p0:p1])
                _to_velocity(v, res_y, dt[p0:p1, None, None], out=full["v_y"][p0:p1])
            done.append((p0, p1))

    stats = {"plan": {k: plan[k] for k in ("windows", "load_frames", "align", "depth", "max_depth", "source", "peak_host_bytes", "host_budget")},
             "load_s": 0.0, "waited_s": 0.0, "upload_s": 0.0, "launch_s": 0.0, "chunks": 0, "depth_per_chunk": [], "load_s_per_chunk": [],
             "waited_s_per_chunk": []}
    try:
        for (w0, w1), loads in zip(plan["windows"], plan["loads"]):
            stack = resident.ResidentStack(w0, w1 - w0, dim_size, frames.dtype, plan["align"], launch, signal_threshold, projection)
            pieces = [(f0, lazy[f0:f1]) for f0, f1 in loads]

            def load_and_stage(piece, stack=stack):
                # on a loader thread: materialise the piece AND bring it to the device (upload -- or upload + projection -- into its place:
                # PCIe and the host's page handling stay off the thread that launches; the loaded array is released here as well)
                f0, da = piece
                return stack.stage(f0, _values(load_frame_chunk(da)))

            with executor.ChunkPrefetcher(pieces, load_and_stage, depth=plan["depth"]) as loader:
                try:
                    for n, n_staged in loader:
                        pieces[n] = None
                        stack.commit(loads[n][0], n_staged)
                finally:
                    loader.close()
                    st = loader.stats
                    stats["load_s"] += st["load_s"]; stats["waited_s"] += st["waited_s"]; stats["chunks"] += st["chunks"]
                    stats["depth_per_chunk"] += st["depth_per_chunk"]
                    stats["load_s_per_chunk"] += st["load_s_per_chunk"]; stats["waited_s_per_chunk"] += st["waited_s_per_chunk"]
                    stats["depth"], stats["workers"], stats["adaptive"] = st["depth"], st["workers"], st["adaptive"]
            stack.finish()
            stats["upload_s"] += stack.upload_s; stats["launch_s"] += stack.launch_s
            del stack
        if not done:
            raise ValueError("no chunk with at least one frame pair")
        if ens is not None:
            # quirk Q3: `n_frames` is the number of CHUNKS of the reference's plan (ffpiv.py:373), `time[0:1]` that of its LAST chunk (:336)
            n_chunks = len(ref_slices) if ref_slices else len(plan["loads"][0])
            u, v, corr_count = ens.finish(count_min, n_chunks)
    finally:
        executor.LAST_STATS.clear(); executor.LAST_STATS.update({k: (round(v, 6) if isinstance(v, float) else v) for k, v in stats.items()})
        if ens is not None:
            ens.close()
    done.sort()
    whole = done[0][0] == 0 and done[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(done, done[1:]))
    if ens is not None:
        cm, sn = full if whole else tuple(np.concatenate([f[i:j] for i, j in done], axis=0) for f in full)
        t_first = time[ref_slices[-1][0] + 1:ref_slices[-1][0] + 2] if ref_slices else time[1:2]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            cm[:, corr_count < count_min * n_chunks] = np.nan
            corr_max_mean = np.nanmean(cm, axis=0).reshape(-1, n_rows, n_cols)
            s2n_mean = np.nanmean(sn, axis=0).reshape(-1, n_rows, n_cols)
        dt_av = dt.mean()
        data = {"s2n": s2n_mean, "corr": corr_max_mean, "v_x": (u * res_x / dt_av).astype(np.float32), "v_y": (v * res_y / dt_av).astype(np.float32)}
        return _dataset(data, t_first, y, x, frames)
    data = full if whole else {k: np.concatenate([full[k][i:j] for i, j in done], axis=0) for k in names}
    times = [time[i + 1:j + 1] for i, j in done]
    if len(times) == 1:
        t = times[0]
    elif _is_xr(frames):
        t = xr.concat(times, dim="time")
    else:
        t = np.concatenate([np.asarray(tt) for tt in times])
    return _dataset(data, t, y, x, frames)
