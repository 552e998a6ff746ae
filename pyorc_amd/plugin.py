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

``import pyorc_amd`` calls ``install()`` by itself when a ``pyorc`` package can be found (``LSPIV_NO_AUTO_INSTALL=1`` turns that
off); ``uninstall()`` restores the originals.  Recipes then simply say ``velocimetry: get_piv: {engine: hip}``.
"""

from __future__ import annotations

import contextvars
import functools
import importlib
import importlib.util
import os
import sys
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
    @functools.wraps(orig)
    def get_ffpiv(frames, y, x, dt, *args, **kwargs):
        if kwargs.get("engine") == ENGINE or _route_hip.get():
            from . import velocimetry

            kwargs["engine"] = ENGINE
            return velocimetry.get_ffpiv(frames, y, x, dt, *args, **kwargs)
        return orig(frames, y, x, dt, *args, **kwargs)

    get_ffpiv.__lspiv_original__ = orig
    return get_ffpiv


def _wrap_get_piv(orig):
    @functools.wraps(orig)
    def get_piv(self, window_size=None, overlap=None, engine="numba", ensemble_corr=False, **kwargs):
        if engine != ENGINE:
            return orig(self, window_size=window_size, overlap=overlap, engine=engine, ensemble_corr=ensemble_corr, **kwargs)
        # fail before any work if there is no MI355X / no library: the reference raises ValueError for an engine it cannot run
        from . import _lib

        _lib.load()
        _lib.require_device()
        token = _route_hip.set(True)
        try:
            # the reference's own method body; its gate (frames.py:176-177) sees an engine it knows, its call of
            # ffpiv.get_ffpiv (frames.py:186-188) is the wrapped function above, which sees the context variable
            return orig(self, window_size=window_size, overlap=overlap, engine="numba", ensemble_corr=ensemble_corr, **kwargs)
        finally:
            _route_hip.reset(token)

    get_piv.__lspiv_original__ = orig
    return get_piv


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
    _installed.update(ffpiv_mod=ffpiv_mod, frames_cls=frames_cls, velo_pkg=velo_pkg, orig_ffpiv=orig_ffpiv,
                      orig_get_piv=orig_get_piv, reexported=reexported)
    return True


def uninstall() -> None:
    """Put pyorc's own ``get_ffpiv`` / ``Frames.get_piv`` back."""
    if not _installed:
        return
    _installed["ffpiv_mod"].get_ffpiv = _installed["orig_ffpiv"]
    if _installed["reexported"]:
        _installed["velo_pkg"].get_ffpiv = _installed["orig_ffpiv"]
    _installed["frames_cls"].get_piv = _installed["orig_get_piv"]
    _installed.clear()


def auto_install() -> Optional[bool]:
    """What ``import pyorc_amd`` does: install when pyorc is there, silently do nothing when it is not or when it does not
    import (a half-installed pyorc must not break ``import pyorc_amd``); None = switched off by ``LSPIV_NO_AUTO_INSTALL``."""
    if os.environ.get("LSPIV_NO_AUTO_INSTALL"):
        return None
    try:
        return install()
    except Exception:  # pragma: no cover - depends on the environment
        return False
