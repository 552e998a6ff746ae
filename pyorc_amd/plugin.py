"""Run-time registration of ``engine="hip"`` in an installed, unmodified pyorc.

The reference's plug-in seam is the ``engine`` string of ``Frames.get_piv`` (pyorc/api/frames.py:118), gated at
``pyorc/api/frames.py:176-177`` (``if engine not in ["numba", "numpy"]: raise ValueError``) and forwarded to
``ffpiv.get_ffpiv(..., engine=engine, ...)`` (``:186-188``, module attribute looked up at call time, ``:15``).
INTEGRATION.md section 1 shows the five-line diff a maintainer would merge; :func:`install` has the same effect without
touching pyorc's files:

* ``pyorc.velocimetry.ffpiv.get_ffpiv`` (and its re-export ``pyorc.velocimetry.get_ffpiv``) is wrapped: a call with
  ``engine="hip"`` goes to :func:`pyorc_amd.velocimetry.get_ffpiv` (same signature, ffpiv.py:24-42), every other engine to the
  original function;
* ``pyorc.api.frames.Frames.get_piv`` is wrapped: ``engine="hip"`` runs THE REFERENCE'S OWN METHOD BODY -- camera configuration
  copy, window / overlap resolution, ``get_piv_coords``, and after the engine call ``add_xy_coords``, attributes,
  ``set_encoding`` -- with the gate satisfied (the original is entered with ``engine="numba"``) and its one call of
  ``ffpiv.get_ffpiv`` routed to the HIP engine through a context variable.  Whatever version of pyorc is installed keeps its own
  code around the engine call; nothing of it is re-implemented here.

A second seam needs no wrapping at all: ``Frames.project`` looks its method up BY NAME, ``getattr(project, f"project_{method}")``
(pyorc/api/frames.py:254-257), so ``install()`` simply adds :func:`project_hip` to ``pyorc.project`` and
``frames.project(method="hip")`` works in an unmodified pyorc: the same lazy graph as ``project_numpy`` (pyorc/project.py:164-230:
``xr.apply_ufunc(..., dask="parallelized")``), whose blocks run the orthoprojection kernel on whole time chunks instead of
``img_to_ortho`` frame by frame.  It is that graph that executes inside ``load_frame_chunk`` ("large in real
runs"), ahead of the PIV launches (pyorc_amd.executor).

``import pyorc_amd`` arranges for ``install()`` by itself (``LSPIV_NO_AUTO_INSTALL=1`` turns that off): at once when ``pyorc`` is already
imported, otherwise the moment it IS imported (a post-import hook on ``sys.meta_path``; ``import pyorc_amd`` alone never imports pyorc
with its xarray / dask / cv2 / numba -- ranks and worker processes that only need the engine stay light).  A failing installation warns
instead of hiding.  ``uninstall()`` restores the originals.  Recipes then simply say ``velocimetry: get_piv: {engine: hip}``.

Third seam (round 6), between the two: when the stack handed to ``get_ffpiv(engine="hip")`` IS the product of ``project_hip`` -- with
nothing but ``Frames.project``'s ``fillna(0.0)`` (pyorc/api/frames.py:265; a no-op on the kernel's NaN-free output) after it, which is
how pyorc's own service strings the two together (pyorc/service/velocimetry.py:537-538) --, the projected frames never travel: the
CAMERA chunk is loaded and uploaded, the orthoprojection kernel writes into the HBM-resident stack the PIV kernels read
(``pyorc_amd.resident``).  ``project_hip`` registers its graph node, :func:`hip_projection_source` recognises it; anything else between
``project`` and ``get_piv`` takes the generic path (the blocks run the kernel, the float32 frames come back and go up again).
"""

from __future__ import annotations

import contextvars
import functools
import importlib
import importlib.abc
import importlib.util
import os
import sys
import threading
import warnings
from collections import OrderedDict
from typing import Optional

_route_hip: contextvars.ContextVar = contextvars.ContextVar("lspiv_route_hip", default=False)
_installed: dict = {}
ENGINE = "hip"


def pyorc_available() -> bool:
    """Is there a ``pyorc`` package to patch (already imported, or importable)?  Does not import it."""
    if "pyorc" in sys.modules:
        return True
    try:
        return importlib.util.find_spec("pyorc") is not None
    except (ImportError, ValueError):
        return False


def is_installed() -> bool:
    return bool(_installed)


def _wrap_get_ffpiv(orig):
    import inspect

    try:
        sig = inspect.signature(orig)
    except (TypeError, ValueError):
        sig = None

    @functools.wraps(orig)
    def get_ffpiv(frames, y, x, dt, *args, **kwargs):
        engine = kwargs.get("engine")
        if engine is None and sig is not None and args:     # `engine` handed over by position (ffpiv.py:24-42: the 12th parameter)
            try:
                bound = sig.bind_partial(frames, y, x, dt, *args, **kwargs)
                engine = bound.arguments.get("engine")
                if engine == ENGINE or _route_hip.get():
                    first4 = list(sig.parameters)[:4]     # frames, y, x, dt under whatever names
                    args, kwargs = (), {k: v for k, v in bound.arguments.items() if k not in first4}
            except TypeError:
                pass
        if engine == ENGINE or _route_hip.get():
            from . import velocimetry

            kwargs["engine"] = ENGINE
            return velocimetry.get_ffpiv(frames, y, x, dt, *args, **kwargs)
        return orig(frames, y, x, dt, *args, **kwargs)

    get_ffpiv.__lspiv_original__ = orig
    return get_ffpiv


def _wrap_get_piv(orig):
    import inspect

    try:
        sig = inspect.signature(orig)
    except (TypeError, ValueError):
        sig = None

    @functools.wraps(orig)
    def get_piv(self, *args, **kwargs):
        # `engine` wherever the installed pyorc's signature has it (frames.py:114-121: the third parameter today): bound by the
        # ORIGINAL's signature, not by a copy of it that another release may not match
        engine, bound = kwargs.get("engine"), None
        if sig is not None and "engine" in sig.parameters:
            try:
                bound = sig.bind(self, *args, **kwargs)
                engine = bound.arguments.get("engine", sig.parameters["engine"].default)
            except TypeError:
                bound = None      # let the original raise its own TypeError
        if engine != ENGINE:
            return orig(self, *args, **kwargs)
        # fail before any work if there is no MI355X / no library: the reference raises ValueError for an engine it cannot run
        from . import _lib

        _lib.load()
        _lib.require_device()
        if bound is not None:
            bound.arguments["engine"] = "numba"
            call_args, call_kwargs = bound.args, bound.kwargs
        else:
            call_args, call_kwargs = (self,) + args, {**kwargs, "engine": "numba"}
        token = _route_hip.set(True)
        try:
            # the reference's own method body; its gate (frames.py:176-177) sees an engine it knows, its call of
            # ffpiv.get_ffpiv (frames.py:186-188) is the wrapped function above, which sees the context variable
            return orig(*call_args, **call_kwargs)
        finally:
            _route_hip.reset(token)

    get_piv.__lspiv_original__ = orig
    return get_piv


_PLANS: dict = {}          # (ids of the index-map arrays, shapes, device) -> (the arrays, Projection): the blocks of one graph share the arrays
_PLANS_MAX = 4
_PLANS_LOCK = threading.Lock()


def _projection_plan(src_shape, dst_shape, plan_args, device=None):
    """Device-resident plan for these index maps: uploaded once per graph and device, a few graphs kept.  dask's threads call the blocks
    of a graph concurrently, all with the SAME index-map arrays -- their identities are the key (no hashing of megabytes of indices per
    block).  The identities of the ARRAYS, not of the tuple that carries them: a real dask re-creates the tuple for every task (tuples
    are its task syntax), which made round 5's ``id(plan_args)`` key miss on every block -- a plan upload per block; found by
    tests/test_real_dask.py.  An evicted plan is only dropped here; it closes itself when the last block that uses it has returned."""
    from .project import Projection

    maps = tuple(plan_args)
    key = (tuple(id(m) for m in maps), tuple(src_shape), tuple(dst_shape), device)
    with _PLANS_LOCK:
        hit = _PLANS.get(key)
        if hit is not None and len(hit[0]) == len(maps) and all(a is b for a, b in zip(hit[0], maps)):
            return hit[1]
        while len(_PLANS) >= _PLANS_MAX:
            _PLANS.pop(next(iter(_PLANS)))
        plan = Projection(src_shape, dst_shape, *maps)
        _PLANS[key] = (maps, plan)       # holds the arrays: their ids cannot be recycled while the entry lives
        return plan


def _project_block(block, plan_args=None, dst_shape=None, device=None):
    """One dask block of frames, core dimensions last: (..., Hc, Wc) -> (..., Ho, Wo) float32, every leading index (time; rgb when the
    frames carry it) projected by ONE kernel call -- on the device the graph was built for (``hipSetDevice`` is per thread, and dask's
    worker threads are nobody's in particular: without this every rank's blocks would pile onto device 0)."""
    import numpy as np

    from . import executor

    executor.bind_device(device)
    a = np.asarray(block)
    lead, src_shape = a.shape[:-2], a.shape[-2:]
    if a.dtype not in (np.dtype(np.uint8), np.dtype(np.float32), np.dtype(np.float64)):
        a = a.astype(np.float32)
    if a.size == 0:
        return np.zeros(lead + tuple(dst_shape), np.float32)
    plan = _projection_plan(src_shape, dst_shape, plan_args, device)
    out = plan.project_frames(np.ascontiguousarray(a.reshape((-1,) + src_shape)), keep_uint8=False)
    return np.asarray(out, dtype=np.float32).reshape(lead + tuple(dst_shape))


# ---- the graph nodes project_hip produced: what lets get_ffpiv load the CAMERA frames and project them where the PIV kernels read ----
_PROJECTIONS: "OrderedDict[str, dict]" = OrderedDict()     # dask array name of a project_hip result -> its recipe
_PROJECTIONS_MAX = 8
_PASS_NOT = ("invert", "logical_not", "notnull")
_PASS_NAN = ("isnan", "isnull")


def _graph_name(obj):
    data = getattr(obj, "data", None)
    name = getattr(data, "name", None)
    return (name, data) if isinstance(name, str) else (None, None)


def _register_projection(da_proj, source, plan_args, dst_shape, device) -> None:
    name, _ = _graph_name(da_proj)
    if name is None:          # an eager result (no dask): nothing lazy to short-cut
        return
    with _PLANS_LOCK:
        _PROJECTIONS[name] = {"source": source, "plan_args": plan_args, "dst_shape": tuple(int(v) for v in dst_shape), "device": device}
        while len(_PROJECTIONS) > _PROJECTIONS_MAX:
            _PROJECTIONS.popitem(last=False)


def _prefix(name: str) -> str:
    return name.rsplit("-", 1)[0] if "-" in name else name


def _match_projection(name, graph):
    """The registered ``project_hip`` node that ``name`` IS, or is ``fillna`` of: ``where(~isnan(P), P, c)`` -- xarray's
    ``duck_array_ops.fillna`` on a dask array, three element-wise layers ``where`` <- {``invert`` <- ``isnan`` <- P, P} (or
    ``where`` <- {``notnull`` <- P, P}).  Anything else (another filter, a spatial selection, a dtype change, a persisted / optimised
    graph without layer names): None, and the generic path runs."""
    with _PLANS_LOCK:
        if name in _PROJECTIONS:
            return _PROJECTIONS[name]
        known = dict(_PROJECTIONS)
    deps = getattr(graph, "dependencies", None)
    if not isinstance(deps, dict) or _prefix(name) != "where":
        return None
    top = set(deps.get(name, ()))
    hits = [d for d in top if d in known]
    if len(top) != 2 or len(hits) != 1:
        return None
    p, cond = hits[0], next(iter(top - {hits[0]}))
    d1 = set(deps.get(cond, ()))
    if _prefix(cond) == "notnull" and d1 == {p}:
        return known[p]
    if _prefix(cond) in _PASS_NOT and len(d1) == 1:
        inner = next(iter(d1))
        if _prefix(inner) in _PASS_NAN and set(deps.get(inner, ())) == {p}:
            return known[p]
    return None


def hip_projection_source(frames) -> Optional[dict]:
    """``{"source", "plan_args", "dst_shape", "device"}`` when ``frames`` is the product of :func:`project_hip` (module docstring, third
    seam) over a plain ``(time, y, x)`` camera stack of the same length, else None."""
    import numpy as np

    name, data = _graph_name(frames)
    if name is None or not _PROJECTIONS:
        return None
    try:
        hit = _match_projection(name, getattr(data, "dask", None))
        if hit is None:
            return None
        src = hit["source"]
        ok = (len(src.shape) == 3 and len(frames.shape) == 3 and len(src) == len(frames) and tuple(frames.shape[1:]) == hit["dst_shape"]
              and np.dtype(frames.dtype) == np.float32)
        return hit if ok else None
    except Exception:
        return None


def projection_for(handoff: dict):
    """The device-resident :class:`pyorc_amd.project.Projection` of a recognised ``project_hip`` node (shared with its dask blocks)."""
    src = handoff["source"]
    return _projection_plan(tuple(int(v) for v in src.shape[-2:]), handoff["dst_shape"], handoff["plan_args"], handoff["device"])


def project_hip(da, cc, x, y, z, reducer="mean"):
    """``pyorc.project.project_hip``: the signature and the result of ``project_numpy`` (pyorc/project.py:164-230) with the gather on the
    MI355X.  The index maps come from the camera configuration exactly as there (``cc.map_idx_img_ortho``,
    ``cc.map_mean_idx_img_ortho`` for ``reducer="mean"``); the frames stay a lazy DataArray -- ``xr.apply_ufunc`` with
    ``dask="parallelized"`` over the time chunks, core dimensions ``(y, x) -> (new_y, new_x)`` --, each block is one call of the
    projection kernel (``pyorc_amd.project.Projection``: bit-identical to ``img_to_ortho``, NaN-free), float32 out (the reference
    declares the frames' dtype and delivers float64; the PIV kernels read float32)."""
    import numpy as np
    import xarray as xr

    from . import _lib

    _lib.load()
    _lib.require_device()   # fail when the graph is BUILT, not in the middle of a dask computation
    idx_img, idx_ortho = cc.map_idx_img_ortho(x, y, z)
    if reducer == "mean":
        src_idx, uidx, norm_idx = cc.map_mean_idx_img_ortho(x, y, z)
    else:
        src_idx = uidx = norm_idx = None
    dst_shape = (len(y), len(x))
    from . import executor

    device = executor.current_device()      # the graph's blocks run on whatever threads dask has: they select THIS device first
    plan_args = (idx_img, idx_ortho, src_idx, uidx, norm_idx)
    da_proj = xr.apply_ufunc(
        _project_block,
        da,
        kwargs={"plan_args": plan_args, "dst_shape": dst_shape, "device": device},
        input_core_dims=[["y", "x"]],
        output_core_dims=[["new_y", "new_x"]],
        dask_gufunc_kwargs={"output_sizes": {"new_y": len(y), "new_x": len(x)}},
        output_dtypes=[np.float32],
        vectorize=False,                 # a block arrives whole, (time_chunk, y, x): one kernel call per block
        exclude_dims=set(("y", "x")),
        dask="parallelized",
        keep_attrs=True,
    ).rename({"new_y": "y", "new_x": "x"})
    da_proj["y"] = y
    da_proj["x"] = x
    _register_projection(da_proj, da, plan_args, dst_shape, device)
    return da_proj


def install(pyorc_module=None) -> bool:
    """Make ``frames.get_piv(engine="hip")`` and ``get_ffpiv(engine="hip")`` of the installed pyorc run on the MI355X.

    Returns True when pyorc was found and is (now) patched, False when there is no pyorc to patch.  Idempotent.
    ``pyorc_module``: the imported ``pyorc`` package (default: ``import pyorc``).
    """
    if _installed:
        return True
    if pyorc_module is None:
        if not pyorc_available():
            return False
        pyorc_module = importlib.import_module("pyorc")
    name = pyorc_module.__name__
    ffpiv_mod = importlib.import_module(name + ".velocimetry.ffpiv")
    frames_mod = importlib.import_module(name + ".api.frames")
    velo_pkg = importlib.import_module(name + ".velocimetry")
    frames_cls = frames_mod.Frames
    orig_ffpiv = ffpiv_mod.get_ffpiv
    orig_get_piv = frames_cls.get_piv
    wrapped = _wrap_get_ffpiv(orig_ffpiv)
    ffpiv_mod.get_ffpiv = wrapped
    reexported = getattr(velo_pkg, "get_ffpiv", None) is orig_ffpiv
    if reexported:
        velo_pkg.get_ffpiv = wrapped
    frames_cls.get_piv = _wrap_get_piv(orig_get_piv)
    project_mod = None
    try:   # Frames.project(method="hip"): found by name (frames.py:254-257)
        project_mod = importlib.import_module(name + ".project")
        if not hasattr(project_mod, "project_hip"):
            project_mod.project_hip = project_hip
        else:
            project_mod = None   # somebody else's: leave it
    except ImportError:
        project_mod = None
    _installed.update(project_mod=project_mod, ffpiv_mod=ffpiv_mod, frames_cls=frames_cls, velo_pkg=velo_pkg, orig_ffpiv=orig_ffpiv,
                      orig_get_piv=orig_get_piv, reexported=reexported)
    return True


def uninstall() -> None:
    """Put pyorc's own ``get_ffpiv`` / ``Frames.get_piv`` back; drop the cached projection plans and the registered graph nodes."""
    if not _installed:
        with _PLANS_LOCK:
            _PLANS.clear()
            _PROJECTIONS.clear()
        _remove_hook()
        return
    _installed["ffpiv_mod"].get_ffpiv = _installed["orig_ffpiv"]
    if _installed["reexported"]:
        _installed["velo_pkg"].get_ffpiv = _installed["orig_ffpiv"]
    _installed["frames_cls"].get_piv = _installed["orig_get_piv"]
    if _installed.get("project_mod") is not None and getattr(_installed["project_mod"], "project_hip", None) is project_hip:
        del _installed["project_mod"].project_hip
    with _PLANS_LOCK:
        _PLANS.clear()
        _PROJECTIONS.clear()
    _installed.clear()
    _remove_hook()


class _InstallAfterImport(importlib.abc.MetaPathFinder):
    """Post-import hook: lets the regular finders locate ``pyorc``, wraps the loader's ``exec_module`` and patches the package right
    after its own ``__init__`` has run.  Removes itself after the first use, whatever the outcome."""

    def __init__(self):
        self._busy = threading.local()

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "pyorc" or getattr(self._busy, "on", False):
            return None
        self._busy.on = True
        try:
            spec = importlib.util.find_spec(fullname)
        except (ImportError, ValueError):
            spec = None
        finally:
            self._busy.on = False
        loader = getattr(spec, "loader", None)
        if spec is None or loader is None or not hasattr(loader, "exec_module"):
            return None
        exec_module = loader.exec_module

        def exec_and_install(module):
            _remove_hook()
            try:
                exec_module(module)
            finally:
                try:
                    loader.exec_module = exec_module
                except Exception:
                    pass
            _install_or_warn(module)

        try:
            loader.exec_module = exec_and_install
        except Exception:       # a loader that does not take attributes: stay out of the way
            return None
        return spec


_hook: Optional[_InstallAfterImport] = None


def _remove_hook() -> None:
    global _hook
    if _hook is not None:
        try:
            sys.meta_path.remove(_hook)
        except ValueError:
            pass
        _hook = None


def _install_or_warn(module=None) -> bool:
    try:
        return install(module)
    except Exception as exc:
        warnings.warn(f"pyorc_amd could not register engine='hip' in pyorc ({type(exc).__name__}: {exc}); "
                      "get_piv(engine='hip') will raise pyorc's 'engine does not exist' -- call pyorc_amd.install() to see the error",
                      RuntimeWarning, stacklevel=2)
        return False


def auto_install():
    """What ``import pyorc_amd`` does: None = switched off by ``LSPIV_NO_AUTO_INSTALL``; pyorc already imported -> ``install()`` now (a
    failure WARNS, it is not swallowed: ADVICE r05); pyorc importable but not imported -> ``"deferred"``: a post-import hook installs
    the moment somebody imports it (``import pyorc_amd`` itself never pulls in pyorc's xarray / dask / cv2 / numba); no pyorc -> False."""
    global _hook
    if os.environ.get("LSPIV_NO_AUTO_INSTALL"):
        return None
    if "pyorc" in sys.modules:
        return _install_or_warn()
    if not pyorc_available():
        return False
    if _hook is None:
        _hook = _InstallAfterImport()
        sys.meta_path.insert(0, _hook)
    return "deferred"
