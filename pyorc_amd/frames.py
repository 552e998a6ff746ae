"""Mirror of ``Frames.get_piv`` / ``Frames.get_piv_coords`` (pyorc/api/frames.py:47-197) for ``engine="hip"``.

Two ways in:

* ``get_piv(frames, ...)`` -- a function with the accessor's parameters.  ``frames`` is either an
  ``xr.DataArray`` produced by pyorc (then camera configuration, resolution, coordinates and
  attributes are taken from it exactly like the reference does, and an ``xr.Dataset`` comes back), or
  a plain ``(T, H, W)`` array -- or a ``pyorc_amd.device.DeviceFrames`` stack that already lives in HBM, e.g. the
  output of ``filters.normalize`` -> ``Projection.project_frames`` on device stacks: the recipe then never bounces
  through host float64 -- plus ``time`` / ``resolution`` keywords (then a ``PivResult`` dict comes back; this is what
  the tests and the benchmark use, since neither xarray nor pyorc exist on the GPU box).
* the two-line patch of INTEGRATION.md, which makes ``frames.frames.get_piv(engine="hip")`` of an
  unmodified pyorc call :func:`pyorc_amd.velocimetry.get_ffpiv`.

Parameter resolution follows the reference line by line: window -> (wy, wx) rounded to even
(:167), search area = window (:168), default overlap from the UN-rounded size (:171, quirk Q6),
engine check raising the same ValueError (:176-177).
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import velocimetry, window

ENGINES = ["hip"]


def resolve_window(window_size, overlap=None) -> Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]]:
    """(window_size, search_area_size, overlap) as pyorc/api/frames.py:159-171 derives them."""
    if window_size is None:
        raise ValueError("window_size is required when frames carry no camera configuration")
    ws = 2 * (window_size,) if isinstance(window_size, (int, np.integer)) else tuple(window_size)
    ws_even = window.round_to_even(ws)
    if overlap is None:
        if isinstance(window_size, (int, np.integer)):
            overlap = 2 * (int(round(window_size) / 2),)
        else:
            overlap = tuple(int(round(w) / 2) for w in window_size)
    return ws_even, ws_even, tuple(int(o) for o in overlap)


def get_piv_coords(dim_size, window_size, search_area_size, overlap, x=None, y=None):
    """Result axes: ``x[cols]``, ``y[rows]`` at the window centres (pyorc/api/frames.py:85-92, helpers.py:166-167)."""
    cols, rows = window.get_rect_coordinates(dim_size=dim_size, window_size=window_size,
                                             search_area_size=search_area_size, overlap=overlap)
    x = np.arange(dim_size[1]) if x is None else np.asarray(x)
    y = np.arange(dim_size[0]) if y is None else np.asarray(y)
    return {"y": y[rows], "x": x[cols]}, {"cols": cols, "rows": rows}


def get_piv(frames, window_size=None, overlap=None, engine: str = "hip", ensemble_corr: bool = False,
            time=None, resolution: Optional[float] = None, **kwargs):
    """PIV on projected frames with the MI355X engine; parameters of ``Frames.get_piv`` (frames.py:114-121).

    Extra keywords for plain arrays: ``time`` (T,) seconds (default ``arange(T)``), ``resolution`` metres per
    pixel (default 1.0 => velocities in px/s).  ``**kwargs`` are forwarded to ``get_ffpiv`` (``chunksize``,
    ``memory_factor``, ``corr_min``, ``s2n_min``, ``count_min``, ``signal_threshold``).
    """
    if engine not in ENGINES:
        raise ValueError(f"Selected PIV engine {engine} does not exist.")
    is_xr = velocimetry._is_xr(frames)
    camera_config = None
    if is_xr and hasattr(frames, "frames") and hasattr(frames.frames, "camera_config"):
        import copy

        camera_config = copy.deepcopy(frames.frames.camera_config)
        if window_size is not None:
            camera_config.window_size = window_size
        window_size = camera_config.window_size
        resolution = camera_config.resolution if resolution is None else resolution
    ws, sa, ov = resolve_window(window_size, overlap)
    if is_xr:
        t = frames["time"]
        dt = t.diff(dim="time")
        xs, ys = frames["x"].values, frames["y"].values
    else:
        # DeviceFrames: HBM-resident stack, used as it is; a lazy stack (anything with ``load()``: time slices materialise on demand,
        # like the dask-backed DataArray of the xarray branch) is left lazy -- get_ffpiv loads it chunk by chunk, ahead of the launches
        if not velocimetry.is_device(frames) and not hasattr(frames, "load"):
            frames = np.asarray(frames)
        t = np.arange(frames.shape[0], dtype=np.float64) if time is None else np.asarray(time, dtype=np.float64)
        dt = np.diff(t)
        xs = ys = None
    res = 1.0 if resolution is None else float(resolution)
    coords, _ = get_piv_coords(tuple(frames[0].shape), ws, sa, ov, xs, ys)
    ds = velocimetry.get_ffpiv(frames, coords["y"], coords["x"], dt, engine=engine, ensemble_corr=ensemble_corr,
                               search_area_size=sa, window_size=ws, overlap=ov, res_x=res, res_y=res,
                               **({} if is_xr else {"time": t}), **kwargs)
    if is_xr and camera_config is not None:
        # the tail of the reference accessor (frames.py:190-196): 2-D coordinates, attributes, encoding
        from pyorc import const  # only reachable inside a pyorc installation

        _, mesh_coords = frames.frames.get_piv_coords(ws, sa, ov)
        ds = ds.velocimetry.add_xy_coords(mesh_coords, coords, {**const.PERSPECTIVE_ATTRS, **const.GEOGRAPHICAL_ATTRS})
        ds.attrs = frames.attrs
        ds.attrs.update(camera_config=camera_config.to_json())
        ds.velocimetry.set_encoding()
    return ds


def encode_int16(a: np.ndarray, scale: float = 0.01, fill: int = -9999) -> np.ndarray:
    """netCDF packing of the result variables (pyorc/const.py:80): int16, scale 0.01, fill -9999 -- on the GPU
    (``lspiv_pack_int16``; no host implementation: without a device this raises like every other entry point)."""
    from .project import pack_int16

    return pack_int16(a, scale, fill)
