"""pyorc_amd.install(): engine="hip" inside an installed, UNMODIFIED pyorc (pyorc_amd/plugin.py).

pyorc is not installed in the build image, so the test builds a double of the package with the members the patch touches
(pyorc.api.frames.Frames.get_piv with the gate of pyorc/api/frames.py:176-177 and the call of :186-188; pyorc.velocimetry.ffpiv.get_ffpiv
with the signature of ffpiv.py:24-42; the re-export of pyorc/velocimetry/__init__.py) on the xarray double of tests/fake_xarray.py.
The GPU call is replaced by the oracle."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_pyorc(fake_xarray, calls):
    """A package `pyorc` as far as the engine seam goes."""
    pyorc = types.ModuleType("pyorc"); pyorc.__path__ = []
    api = types.ModuleType("pyorc.api"); api.__path__ = []
    velo = types.ModuleType("pyorc.velocimetry"); velo.__path__ = []
    ffpiv = types.ModuleType("pyorc.velocimetry.ffpiv")
    frames = types.ModuleType("pyorc.api.frames")

    def get_ffpiv(frames_, y, x, dt, window_size, overlap, search_area_size, res_y, res_x, chunksize=None, memory_factor=4,
                  engine="numba", ensemble_corr=False, corr_min=0.2, s2n_min=3, count_min=0.2, signal_threshold=None):
        calls.append(("cpu_get_ffpiv", engine))
        ds = fake_xarray.Dataset({}, coords={"y": y, "x": x})
        ds.cpu = True
        return ds

    ffpiv.get_ffpiv = get_ffpiv
    velo.get_ffpiv = get_ffpiv          # pyorc/velocimetry/__init__.py: from .ffpiv import get_ffpiv
    velo.ffpiv = ffpiv

    class CameraConfig:
        window_size, resolution = 32, 0.02

    class Frames:
        def __init__(self, obj):
            self._obj, self.camera_config = obj, CameraConfig()

        def get_piv(self, window_size=None, overlap=None, engine="numba", ensemble_corr=False, **kwargs):
            calls.append(("Frames.get_piv", engine))
            from pyorc_amd import frames as F   # only for the grid arithmetic of this double

            ws, sa, ov = F.resolve_window(self.camera_config.window_size if window_size is None else window_size, overlap)
            coords, _ = F.get_piv_coords(tuple(self._obj[0].shape), ws, sa, ov, self._obj["x"].values, self._obj["y"].values)
            dt = self._obj["time"].diff(dim="time")
            if engine not in ["numba", "numpy"]:
                raise ValueError(f"Selected PIV engine {engine} does not exist.")
            kwargs = {**kwargs, "search_area_size": sa, "window_size": ws, "overlap": ov, "res_x": self.camera_config.resolution,
                      "res_y": self.camera_config.resolution}
            ds = ffpiv.get_ffpiv(self._obj, coords["y"], coords["x"], dt, engine=engine, ensemble_corr=ensemble_corr, **kwargs)
            ds.attrs = dict(self._obj.attrs, tail_of_the_reference_method=True)
            return ds

    frames.Frames, frames.ffpiv = Frames, ffpiv
    pyorc.api, pyorc.velocimetry, api.frames = api, velo, frames
    return {"pyorc": pyorc, "pyorc.api": api, "pyorc.api.frames": frames, "pyorc.velocimetry": velo, "pyorc.velocimetry.ffpiv": ffpiv}


def test_install_registers_the_hip_engine_in_an_unmodified_pyorc(monkeypatch):
    from oracle import c_oracle
    from pyorc_amd import _lib, plugin
    from pyorc_amd.synth import particle_stack
    from tests import fake_xarray

    calls = []
    mods = _fake_pyorc(fake_xarray, calls)
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    import pyorc_amd.frames as F
    import pyorc_amd.velocimetry as V

    for mod in (V, F):
        importlib.reload(mod)
    plugin.uninstall()
    try:
        monkeypatch.setattr(V.piv, "piv_pairs", __import__("tests.doubles", fromlist=["x"]).oracle_piv_pairs)
        monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
        monkeypatch.setattr(_lib, "require_device", lambda: None)
        fr = particle_stack(6, 96, 128, seed=9)
        t = np.arange(6) / 25.0
        da = fake_xarray.DataArray(fr, ("time", "y", "x"), {"time": t, "y": np.arange(96)[::-1] * 0.02, "x": np.arange(128) * 0.02}, attrs={"h_a": 0.7})
        acc = mods["pyorc.api.frames"].Frames(da)
        with pytest.raises(ValueError, match="Selected PIV engine hip does not exist"):
            acc.get_piv(engine="hip")                                   # the unpatched gate, pyorc/api/frames.py:176-177
        assert plugin.pyorc_available() and plugin.install() and plugin.is_installed() and plugin.install()
        calls.clear()
        ds = acc.get_piv(engine="hip", chunksize=3)
        assert calls == [("Frames.get_piv", "numba")]                  # the reference's own method body ran; its CPU engine did not
        assert ds.attrs == {"h_a": 0.7, "tail_of_the_reference_method": True} and not hasattr(ds, "cpu")
        ref = F.get_piv(fr, 32, time=t, resolution=0.02)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(ds[k].values, ref[k], equal_nan=True)
        assert np.allclose(np.asarray(ds["time"].values, dtype=np.float64), t[1:])
        # other engines: untouched
        calls.clear()
        assert acc.get_piv(engine="numba").cpu and calls == [("Frames.get_piv", "numba"), ("cpu_get_ffpiv", "numba")]
        with pytest.raises(ValueError, match="Selected PIV engine openpiv does not exist"):
            acc.get_piv(engine="openpiv")
        # the wrapper one level down (and its re-export), called the way frames.py:186-188 calls it
        coords, _ = F.get_piv_coords((96, 128), (32, 32), (32, 32), (16, 16), da["x"].values, da["y"].values)
        kw = dict(window_size=(32, 32), overlap=(16, 16), search_area_size=(32, 32), res_x=0.02, res_y=0.02)
        for fn in (mods["pyorc.velocimetry.ffpiv"].get_ffpiv, mods["pyorc.velocimetry"].get_ffpiv):
            d2 = fn(da, coords["y"], coords["x"], da["time"].diff(dim="time"), engine="hip", **kw)
            assert np.array_equal(d2["v_x"].values, ref["v_x"], equal_nan=True)
            calls.clear()
            assert fn(da, coords["y"], coords["x"], da["time"].diff(dim="time"), engine="numpy", **kw).cpu and calls == [("cpu_get_ffpiv", "numpy")]
        # ... and with every argument by position, `engine` being the 12th (ffpiv.py:24-42)
        d3 = mods["pyorc.velocimetry.ffpiv"].get_ffpiv(da, coords["y"], coords["x"], da["time"].diff(dim="time"), (32, 32), (16, 16), (32, 32), 0.02, 0.02,
                                                       None, 4, "hip")
        assert np.array_equal(d3["v_x"].values, ref["v_x"], equal_nan=True)
        # a failing hip call leaves no routing behind: the next numba call reaches the CPU engine
        monkeypatch.setattr(V.piv, "piv_pairs", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("device lost")))
        with pytest.raises(RuntimeError, match="device lost"):
            acc.get_piv(engine="hip")
        calls.clear()
        assert acc.get_piv(engine="numba").cpu and ("cpu_get_ffpiv", "numba") in calls
        # no device: engine="hip" fails loudly before any work, other engines are unaffected
        monkeypatch.setattr(_lib, "require_device", lambda: (_ for _ in ()).throw(_lib.LspivError(-6, "no gfx950 device")))
        with pytest.raises(_lib.LspivError):
            acc.get_piv(engine="hip")
        plugin.uninstall()
        assert not plugin.is_installed() and mods["pyorc.velocimetry"].get_ffpiv is mods["pyorc.velocimetry.ffpiv"].get_ffpiv
        with pytest.raises(ValueError, match="Selected PIV engine hip does not exist"):
            acc.get_piv(engine="hip")
    finally:
        plugin.uninstall()
        monkeypatch.undo()
        for mod in (V, F):
            importlib.reload(mod)


def test_auto_install_is_silent_without_pyorc_and_can_be_switched_off(monkeypatch):
    from pyorc_amd import plugin

    plugin.uninstall()
    assert not plugin.pyorc_available()              # the build image has no pyorc: importing pyorc_amd patched nothing
    assert plugin.install() is False and plugin.auto_install() is False and not plugin.is_installed()
    monkeypatch.setenv("LSPIV_NO_AUTO_INSTALL", "1")
    assert plugin.auto_install() is None


class _ProjectDoubles:
    """pyorc.project + the camera-configuration members Frames.project touches around the method lookup of pyorc/api/frames.py:254-257."""

    def __init__(self, fake_xarray, maps, dst_shape):
        self.maps, self.dst_shape, self.calls = maps, dst_shape, []
        project = types.ModuleType("pyorc.project")
        project.xr = fake_xarray

        def project_numpy(da, cc, x, y, z, reducer="mean"):      # only so that the module looks like the reference's
            raise AssertionError("the CPU projection must not run")

        project.project_numpy = project_numpy
        self.module = project
        outer = self

        class CameraConfig:
            def map_idx_img_ortho(self, x, y, z):
                outer.calls.append(("map_idx_img_ortho", len(x), len(y), z))
                return outer.maps[0], outer.maps[1]

            def map_mean_idx_img_ortho(self, x, y, z):
                outer.calls.append(("map_mean_idx_img_ortho", len(x), len(y), z))
                return outer.maps[2], outer.maps[3], outer.maps[4]

        self.cc = CameraConfig()

    def frames_project(self, da, method, reducer="mean"):
        """The lookup-by-name of Frames.project (pyorc/api/frames.py:254-258) and the fillna that follows it."""
        y, x = np.arange(self.dst_shape[0])[::-1] * 0.01, np.arange(self.dst_shape[1]) * 0.01
        if not hasattr(self.module, f"project_{method}"):
            raise ValueError(f"Selected projection method {method} does not exist.")
        proj_method = getattr(self.module, f"project_{method}")
        return proj_method(da, self.cc, x, y, 1.25, reducer), x, y


def test_install_adds_project_hip_found_by_name(monkeypatch):
    """``Frames.project(method="hip")`` in an unmodified pyorc: the method is looked up by name in pyorc.project (frames.py:254-257),
    install() puts ``project_hip`` there.  It builds the same apply_ufunc graph as project_numpy (an eager double here, cut in blocks
    like dask's time chunks); every block is one call of the projection plan.  CPU: the plan is replaced by the numpy oracle."""
    from oracle import project_oracle as pro
    from pyorc_amd import _lib, plugin
    from pyorc_amd import project as P
    from pyorc_amd.synth import particle_stack, projection_maps
    from tests import fake_xarray

    src, dst = (96, 128), (72, 100)
    maps = projection_maps(src, dst, tilt=0.2, seed=4)      # idx_img, idx_ortho, src_idx, uidx, norm_idx
    dbl = _ProjectDoubles(fake_xarray, maps, dst)
    calls = []
    mods = _fake_pyorc(fake_xarray, calls)
    mods["pyorc.project"] = dbl.module
    mods["pyorc"].project = dbl.module
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    made = []

    class OraclePlan:
        def __init__(self, src_shape, dst_shape, *m):
            made.append(self)
            self.src_shape, self.dst_shape, self.m, self.blocks = tuple(src_shape), tuple(dst_shape), m, []

        def project_frames(self, frames, keep_uint8=None):
            self.blocks.append(frames.shape)
            assert keep_uint8 is False and frames.flags.c_contiguous
            return pro.project_frames(frames, self.dst_shape, *self.m).astype(np.float32)

        def close(self):
            pass

    monkeypatch.setattr(P, "Projection", OraclePlan)
    monkeypatch.setattr(_lib, "require_device", lambda: None)
    plugin.uninstall()
    try:
        cam = particle_stack(10, src[0], src[1], seed=12)
        da = fake_xarray.DataArray(cam, ("time", "y", "x"), {"time": np.arange(10) / 25.0}, attrs={"h_a": 1.0, "_blocks": 3})
        with pytest.raises(ValueError, match="Selected projection method hip does not exist"):
            dbl.frames_project(da, "hip")
        assert plugin.install() and dbl.module.project_hip is plugin.project_hip
        out, x, y = dbl.frames_project(da, "hip")
        assert out.dims == ("time", "y", "x") and out.values.shape == (10,) + dst and out.values.dtype == np.float32
        assert np.array_equal(out["x"].values, x) and np.array_equal(out["y"].values, y) and out.attrs["h_a"] == 1.0
        assert np.array_equal(out.values.astype(np.float64), pro.project_frames(cam, dst, *maps))
        assert len(made) == 1 and made[0].blocks == [(4, 96, 128), (3, 96, 128), (3, 96, 128)]     # one plan per graph, one call per block
        assert [c[0] for c in dbl.calls] == ["map_idx_img_ortho", "map_mean_idx_img_ortho"] and dbl.calls[0][1:] == (100, 72, 1.25)
        # another reducer: nearest neighbour only, no group-mean maps asked for
        dbl.calls.clear()
        out2, _, _ = dbl.frames_project(da, "hip", reducer="max")
        assert [c[0] for c in dbl.calls] == ["map_idx_img_ortho"] and len(made) == 2 and made[1].m[2:] == (None, None, None)
        assert np.array_equal(out2.values.astype(np.float64), pro.project_frames(cam, dst, maps[0], maps[1]))
        # frames with a trailing rgb axis: core dimensions are (y, x), rgb is a loop dimension like time
        rgb = fake_xarray.DataArray(np.stack([cam, cam // 2, cam // 3], axis=-1), ("time", "y", "x", "rgb"), {"time": np.arange(10) / 25.0})
        out3, _, _ = dbl.frames_project(rgb, "hip")
        assert out3.dims == ("time", "rgb", "y", "x") and np.array_equal(out3.values[:, 1].astype(np.float64), pro.project_frames(cam // 2, dst, *maps))
        plugin.uninstall()
        assert not hasattr(dbl.module, "project_hip")
    finally:
        plugin.uninstall()
        monkeypatch.undo()


REF_FRAMES = "/root/reference/pyorc/api/frames.py"


@pytest.mark.skipif(not os.path.exists(REF_FRAMES), reason="reference checkout not present (GPU box)")
def test_the_reference_has_the_seams_the_plugin_patches():
    """The plug-in assumptions of pyorc_amd/plugin.py read off the reference's SOURCE (parsed, not imported: pyorc's dependencies are
    not installable here): Frames.get_piv's parameters, the engine gate, the one call of ffpiv.get_ffpiv with `engine=` as a keyword
    through the module attribute, pyorc.velocimetry's re-export, and Frames.project's lookup of `project_<method>` by name with the
    (da, cc, x, y, z, reducer) call."""
    import ast

    tree = ast.parse(open(REF_FRAMES).read())
    frames_cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Frames")
    meth = {n.name: n for n in frames_cls.body if isinstance(n, ast.FunctionDef)}
    # get_piv(self, window_size=None, overlap=None, engine="numba", ensemble_corr=False, **kwargs)
    gp = meth["get_piv"]
    assert [a.arg for a in gp.args.args] == ["self", "window_size", "overlap", "engine", "ensemble_corr"] and gp.args.kwarg.arg == "kwargs"
    assert [ast.literal_eval(d) for d in gp.args.defaults] == [None, None, "numba", False]
    src = ast.get_source_segment(open(REF_FRAMES).read(), gp)
    assert 'if engine not in ["numba", "numpy"]:' in src and "raise ValueError" in src            # the gate the wrapper satisfies
    calls = [n for n in ast.walk(gp) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "get_ffpiv"]
    assert len(calls) == 1 and isinstance(calls[0].func.value, ast.Name) and calls[0].func.value.id == "ffpiv"   # looked up on the module at call time
    kw = {k.arg for k in calls[0].keywords}
    assert "engine" in kw and "ensemble_corr" in kw and None in kw and len(calls[0].args) == 4     # (self._obj, y, x, dt, engine=..., **kwargs)
    imports = [n for n in tree.body if isinstance(n, ast.ImportFrom)]
    assert any(n.module == "pyorc.velocimetry" and any(a.name == "ffpiv" for a in n.names) for n in imports)
    assert any(n.module == "pyorc" and any(a.name == "project" for a in n.names) for n in imports)
    # Frames.project(method=...): getattr(project, f"project_{method}") and proj_method(self._obj, cc, x, y, z, reducer)
    pj = ast.get_source_segment(open(REF_FRAMES).read(), meth["project"])
    assert 'getattr(project, f"project_{method}")' in pj and "proj_method(self._obj, cc, x, y, z, reducer)" in pj
    # pyorc/velocimetry/__init__.py re-exports get_ffpiv; ffpiv.get_ffpiv's parameters are the ones pyorc_amd.velocimetry.get_ffpiv mirrors
    init = open("/root/reference/pyorc/velocimetry/__init__.py").read()
    assert "from .ffpiv import get_ffpiv" in init
    ftree = ast.parse(open("/root/reference/pyorc/velocimetry/ffpiv.py").read())
    ref_fn = next(n for n in ftree.body if isinstance(n, ast.FunctionDef) and n.name == "get_ffpiv")
    import inspect

    from pyorc_amd import velocimetry

    ours = list(inspect.signature(velocimetry.get_ffpiv).parameters)
    theirs = [a.arg for a in ref_fn.args.args]
    assert ours[:len(theirs)] == theirs and ours[len(theirs):] == ["time", "prefetch"]           # same names, same order; two extensions at the end
    ref_defaults = dict(zip(theirs[-len(ref_fn.args.defaults):], [ast.literal_eval(d) for d in ref_fn.args.defaults]))
    sig = inspect.signature(velocimetry.get_ffpiv).parameters
    for name, val in ref_defaults.items():
        if name != "engine":
            assert sig[name].default == val, name
    # project_numpy's signature is the one project_hip takes
    ptree = ast.parse(open("/root/reference/pyorc/project.py").read())
    pn = next(n for n in ptree.body if isinstance(n, ast.FunctionDef) and n.name == "project_numpy")
    from pyorc_amd import plugin

    assert [a.arg for a in pn.args.args] == list(inspect.signature(plugin.project_hip).parameters)


@pytest.mark.skipif(not os.path.exists(REF_FRAMES), reason="reference checkout not present (GPU box)")
def test_messages_and_planner_constants_are_the_references():
    """What a user SEES must be the reference's: the OverflowError / UserWarning texts of the chunk planner (ffpiv.py:132-139), the
    ValueError of an unknown engine (frames.py:177), the planner's constants (memory_factor 4, chunksize floor 5) and the int16 encoding
    of the result variables (const.py) -- compared with the reference's source text (parsed, not imported)."""
    import ast

    from pyorc_amd import frames as F, velocimetry as V

    ffpiv_src = open("/root/reference/pyorc/velocimetry/ffpiv.py").read()
    ref_fn = next(n for n in ast.parse(ffpiv_src).body if isinstance(n, ast.FunctionDef) and n.name == "get_ffpiv")

    def norm(t):   # the reference builds its messages from adjacent (f-)string literals; compare the fixed words
        return " ".join(t.replace("{chunks}", " ").replace("{chunksize}", " ").replace("{avail_mem}", " ").replace("{engine}", " ").split())

    ref_text = norm(" ".join(n.value for n in ast.walk(ref_fn) if isinstance(n, ast.Constant) and isinstance(n.value, str) and len(n.value) > 12
                             and n is not ast.get_docstring(ref_fn)))
    for ours in (V.CHUNK_SIZE_ERROR, V.CHUNK_SIZE_WARNING):
        words = norm(ours).split()
        # every run of five consecutive words of our message occurs in the reference's literals
        assert all(" ".join(words[i:i + 5]) in ref_text for i in range(0, len(words) - 5, 5)), ours
    frames_src = open(REF_FRAMES).read()
    assert 'f"Selected PIV engine {engine} does not exist."' in frames_src
    with pytest.raises(ValueError, match="Selected PIV engine numba does not exist."):
        F.get_piv(np.zeros((2, 64, 64), np.uint8), 32, engine="numba")
    assert "memory_factor: float = 4" in ffpiv_src and "chunksize <= 5" in ffpiv_src and "chunksize = 5" in ffpiv_src
    const_src = open("/root/reference/pyorc/const.py").read()
    assert '"scale_factor": 0.01' in const_src and "-9999" in const_src and "int16" in const_src
