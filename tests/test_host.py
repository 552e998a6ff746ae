"""CPU tests of the host side: C-ABI surface, window grid, chunk planner, parameter resolution, loud failure
without a GPU.  No compute entry point is exercised here (there is no GPU and no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import piv_oracle as po
from pyorc_amd import _lib, frames, shard, velocimetry, window

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "lspiv.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lspiv_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/lspiv.h but not exported by liblspiv_hip.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes prototype in pyorc_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.lspiv_abi_version() == 5
    assert b"gfx950" in lib.lspiv_version()


def test_library_contains_gfx950_code_object_only():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"gfx1100", b"sm_90", b"nvptx"):
        assert other not in blob, other  # no second backend, no compatibility layer


@pytest.mark.parametrize("dim,win,ov", [((1080, 1920), (32, 32), (16, 16)), ((1080, 1920), (64, 64), (48, 48)),
                                        ((2160, 3840), (32, 32), (16, 16)), ((785, 875), (32, 32), (16, 16)),
                                        ((475, 371), (10, 10), (5, 5)), ((100, 90), (24, 16), (12, 8)),
                                        ((40, 40), (26, 26), (13, 13))])
def test_grid_matches_oracle(dim, win, ov):
    x, y = window.get_rect_coordinates(dim, win, ov)
    xo, yo = po.get_rect_coordinates(dim, win, ov)
    assert np.array_equal(x, xo) and np.array_equal(y, yo) and x.dtype == np.int64
    assert window.get_array_shape(dim, win, ov) == (len(yo), len(xo))


def test_grid_errors_map_to_status_codes(lib):
    nr, nc = C.c_int64(), C.c_int64()
    assert lib.lspiv_grid_shape(100, 100, 32, 32, 32, 16, C.byref(nr), C.byref(nc)) == _lib.LSPIV_EINVAL
    assert b"overlap" in lib.lspiv_last_error()
    assert lib.lspiv_grid_shape(800, 800, 513, 513, 0, 0, C.byref(nr), C.byref(nc)) == _lib.LSPIV_EUNSUPPORTED
    assert lib.lspiv_grid_shape(200, 200, 128, 128, 64, 64, C.byref(nr), C.byref(nc)) == 0 and (nr.value, nc.value) == (2, 2)
    # windows above 64 px run the LDS-resident DFT kernel (any shape whose two planes fit the 160 KB of a CU)
    assert [lib.lspiv_kernel_kind(w, w) for w in (66, 96, 97, 100, 128)] == [9] * 5
    assert lib.lspiv_kernel_kind(128, 96) == 9 and lib.lspiv_kernel_kind(80, 20) == 9
    assert lib.lspiv_kernel_kind(513, 513) == _lib.LSPIV_EUNSUPPORTED and lib.lspiv_kernel_kind(129, 129) == 10
    assert lib.lspiv_grid_shape(10, 100, 32, 32, 16, 16, C.byref(nr), C.byref(nc)) == 0 and nr.value == 0
    with pytest.raises(_lib.LspivError):
        window.get_rect_coordinates((100, 100), (32, 32), (40, 16))


def test_kernel_dispatch_table(lib):
    assert lib.lspiv_kernel_kind(32, 32) == 1
    assert lib.lspiv_kernel_kind(9, 9) == 4 and lib.lspiv_kernel_kind(15, 15) == 4     # embedded in the 32-point FFT
    assert lib.lspiv_kernel_kind(16, 16) == 6                                            # native 16-point kernels
    assert lib.lspiv_kernel_kind(4, 4) == 7 and lib.lspiv_kernel_kind(7, 7) == 7          # embedded in the 16-point FFT
    assert lib.lspiv_kernel_kind(8, 8) == 6                                              # native 8-point kernels
    assert lib.lspiv_kernel_kind(21, 21) == 5 and lib.lspiv_kernel_kind(25, 25) == 5   # embedded in the 64-point FFT
    assert all(lib.lspiv_kernel_kind(n, n) == 8 for n in range(6, 64, 2) if n not in (8, 16, 32))   # prime-factor FFT kernels
    assert lib.lspiv_kernel_kind(24, 16) == 3 and lib.lspiv_kernel_kind(35, 35) == 3   # direct: non-square, odd 33..63
    assert lib.lspiv_kernel_kind(17, 17) == 3 and lib.lspiv_kernel_kind(19, 19) == 3   # direct: cheaper than 2 x FFT64
    assert lib.lspiv_kernel_kind(40, 32) == 3 and lib.lspiv_kernel_kind(37, 37) == 3   # < 1500 samples: direct
    assert lib.lspiv_kernel_kind(48, 32) == 9 and lib.lspiv_kernel_kind(39, 39) == 9   # from 1500 samples on: DFT passes
    assert lib.lspiv_kernel_kind(41, 41) == 9 and lib.lspiv_kernel_kind(64, 32) == 9 and lib.lspiv_kernel_kind(63, 63) == 9
    assert lib.lspiv_kernel_kind(64, 64) == 2
    assert lib.lspiv_kernel_kind(128, 128) == 9 and lib.lspiv_kernel_kind(256, 256) == 10 and lib.lspiv_kernel_kind(128, 160) == 10   # LDS-resident DFT up to 128 x 128, HBM slots above


def test_required_memory_counts_frames_and_results():
    def rescue_lists(n_tiles):   # the float64 rescue pass's lists on the launch stream (round 3)
        return 256 + max(4096, n_tiles // 4) * 16 + max(1024, n_tiles // 16) * 4

    b = window.required_memory(1001, (1080, 1920), (32, 32), (16, 16), dtype=np.uint8)
    assert b == 1001 * 1080 * 1920 + 4 * 4 * 1000 * 66 * 119 + rescue_lists(1000 * 66 * 119)
    bp = window.required_memory(3, (64, 64), (32, 32), (16, 16), dtype=np.float32, with_planes=True)
    assert bp == 3 * 64 * 64 * 4 + 16 * 2 * 9 + 2 * 9 * 1024 * 4 + rescue_lists(18)
    _lib.set_option("rescue", 0)
    try:
        assert window.required_memory(3, (64, 64), (32, 32), (16, 16), dtype=np.float32) == 3 * 64 * 64 * 4 + 16 * 2 * 9
    finally:
        _lib.set_option("rescue", 1)


def test_no_gpu_means_loud_failure_not_fallback(lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import pyorc_amd

    with pytest.raises(_lib.LspivError) as ei:
        pyorc_amd.piv_pairs(np.zeros((2, 64, 64), np.uint8))
    assert ei.value.code == _lib.LSPIV_ENODEV
    u = np.zeros(9, np.float32)
    rc = lib.lspiv_piv_pairs(_lib.ptr(np.zeros((2, 64, 64), np.uint8)), 0, 2, 64, 64, 32, 32, 16, 16, -1.0,
                             _lib.ptr(u), _lib.ptr(u), _lib.ptr(u), _lib.ptr(u), None)
    assert rc == _lib.LSPIV_ENODEV and b"device" in lib.lspiv_last_error()
    with pytest.raises(_lib.LspivError):
        window.available_memory()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyorc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "piv_oracle" not in txt, f
    code = "import sys; import pyorc_amd, pyorc_amd.velocimetry, pyorc_amd.frames, pyorc_amd.shard; " \
           "assert not any(m.startswith('oracle') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "liblspiv_hip.so"))
    with pytest.raises(_lib.LspivLibraryMissing):
        _lib.load()


def test_resolve_window_matches_reference_rules():
    for ws in (32, 10, 25, 64, (32, 64)):
        assert frames.resolve_window(ws) == po.resolve_piv_args(ws)
    assert frames.resolve_window(32, (8, 24)) == ((32, 32), (32, 32), (8, 24))
    with pytest.raises(ValueError, match="does not exist"):
        frames.get_piv(np.zeros((3, 64, 64), np.uint8), 32, engine="numba")  # same message as frames.py:177
    with pytest.raises(ValueError, match="does not exist"):
        velocimetry.get_ffpiv(np.zeros((3, 64, 64)), [0], [0], [1, 1], (32, 32), (16, 16), (32, 32), 1, 1,
                              engine="openpiv")


@pytest.mark.parametrize("n,req,avail,cs", [(21, 1e9, 1e12, None), (21, 4e9, 1e9, None), (11, 0, 1, 5), (11, 0, 1, 10),
                                            (10, 0, 1, 10), (1001, 9e9, 1e9, None), (7, 0, 1, 2)])
def test_chunk_planner_matches_oracle(n, req, avail, cs):
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, got = velocimetry.plan_chunks(n, req, avail, cs, "hip")
    assert got == po.plan_chunks(n, req, avail, cs)
    assert sum(b - a - 1 for a, b in got) == n - 1  # every pair exactly once


def test_chunk_planner_warning_and_error():
    with pytest.warns(UserWarning, match="Memory availability is poor"):
        cs, _ = velocimetry.plan_chunks(100, 1e12, 1e9, None, "hip")
    assert cs == 5
    with pytest.raises(OverflowError, match="Chunk size"):
        velocimetry.plan_chunks(10, 0, 1, 1, "hip")
    cs, sl = velocimetry.plan_chunks(1001, 0, 1, 1001, "hip", n_win=2**29)  # 32-bit window index per launch
    assert cs == 3 and all(b - a <= 4 for a, b in sl)


def test_shard_blocks_partition_pairs():
    for n, w in [(1000, 8), (8000, 8), (7, 2), (5, 8), (1, 1), (0, 2)]:
        blocks = [shard.pair_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in blocks) - min(b - a for a, b in blocks) <= 1
        for r in range(w):
            a, b = shard.pair_block(n, r, w)
            assert shard.frame_block(n, r, w) == ((a, b + 1) if b > a else (a, a))
    with pytest.raises(ValueError):
        shard.pair_block(10, 2, 2)


def test_int16_encoding_has_no_host_implementation():
    """N4 is a HIP kernel; the Python name is a thin wrapper that needs the device (GPU parity: tests/test_project.py)."""
    a = np.array([0.1234, -1.005, np.nan, 327.0, 1e6])
    assert po.encode_int16(a).tolist() == [12, -100, -9999, 32700, 32767]
    if _lib.device_count() < 1:
        with pytest.raises(_lib.LspivError):
            frames.encode_int16(a)


def test_as_frames_dtype_rules():
    assert _lib.as_frames(np.zeros((2, 4, 4), np.uint8)).dtype == np.uint8
    assert _lib.as_frames(np.zeros((2, 4, 4), np.float32)).dtype == np.float32
    assert _lib.as_frames(np.zeros((2, 4, 4), np.int16)).dtype == np.float32
    assert _lib.as_frames(np.zeros((2, 4, 4), np.int64)).dtype == np.float64
    assert _lib.as_frames(np.zeros((4, 4, 2), np.uint8).transpose(2, 0, 1)).flags.c_contiguous
    with pytest.raises(ValueError):
        _lib.as_frames(np.zeros((4, 4)))


def test_walk_option_overrides_the_environment(lib, monkeypatch):
    """lspiv_set_option("walk", v) wins over LSPIV_WALK; -1 hands control back to the environment (no GPU needed)."""
    import pyorc_amd

    monkeypatch.delenv("LSPIV_WALK", raising=False)
    pyorc_amd.set_option("walk", -1)
    assert pyorc_amd.get_option("walk") == 1
    monkeypatch.setenv("LSPIV_WALK", "0")
    assert pyorc_amd.get_option("walk") == 0
    pyorc_amd.set_option("walk", 31)
    assert pyorc_amd.get_option("walk") == 31
    pyorc_amd.set_option("walk", -1)
    assert pyorc_amd.get_option("walk") == 0
    with pytest.raises(_lib.LspivError):
        pyorc_amd.set_option("no-such-option", 1)
    for name, hi in (("border_peak", 2), ("signal_mode", 1), ("signal_positive", 1)):   # the unpinned ffpiv readings (A5 / A7)
        assert pyorc_amd.get_option(name) == 0
        pyorc_amd.set_option(name, hi)
        assert pyorc_amd.get_option(name) == hi
        with pytest.raises(_lib.LspivError):
            pyorc_amd.set_option(name, hi + 1)
        pyorc_amd.set_option(name, 0)


def test_xarray_branches_of_the_mirrors(monkeypatch):
    """get_piv / get_ffpiv / Mask on DataArray / Dataset inputs (a test double of xarray, tests/fake_xarray.py; the GPU
    call is replaced by the oracle, so this runs on CPU): Dataset out, time = stamp of the 2nd frame, chunks loaded once."""
    import importlib
    import sys

    from oracle import c_oracle
    from pyorc_amd.synth import particle_stack
    from tests import fake_xarray

    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    import pyorc_amd.frames as F
    import pyorc_amd.mask as M
    import pyorc_amd.velocimetry as V

    for mod in (V, F, M):
        importlib.reload(mod)
    try:
        assert V.xr is fake_xarray
        monkeypatch.setattr(V.piv, "piv_pairs", __import__("tests.doubles", fromlist=["x"]).oracle_piv_pairs)
        monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
        fr = particle_stack(7, 96, 128, seed=3)
        t = np.arange(7) / 25.0
        da = fake_xarray.DataArray(fr, ("time", "y", "x"), {"time": t, "y": np.arange(96)[::-1] * 0.02, "x": np.arange(128) * 0.02})
        ds = F.get_piv(da, 32, resolution=0.02, chunksize=3)
        assert isinstance(ds, fake_xarray.Dataset) and set(ds) == {"s2n", "corr", "v_x", "v_y"}
        assert ds["v_x"].dims == ("time", "y", "x") and ds["v_x"].values.shape == (6, 5, 7) and ds["v_x"].values.dtype == np.float32
        assert np.allclose(np.asarray(ds["time"].values, dtype=np.float64), t[1:])
        ref = F.get_piv(fr, 32, time=t, resolution=0.02, chunksize=3)
        for k in ref:
            assert np.array_equal(ds[k].values, ref[k], equal_nan=True)
        # masks on a Dataset: a DataArray comes back, in-place application keeps the Dataset type
        monkeypatch.setattr(M, "run_mask", lambda block, kind, params: getattr(__import__("oracle.mask_oracle", fromlist=["x"]), kind)(block, *params[:1]) if kind in ("corr", "s2n") else None)
        monkeypatch.setattr(M, "apply_mask", lambda block, mask: __import__("oracle.mask_oracle", fromlist=["x"]).apply(block, mask))
        m = M.Mask(ds).corr(tolerance=0.3)
        assert isinstance(m, fake_xarray.DataArray) and m.dims == ("time", "y", "x") and m.values.dtype == bool
        ds["v_x"].attrs["units"] = "m s-1"                       # per-variable attrs / encoding survive an in-place mask
        ds["v_x"].encoding = {"dtype": "int16", "scale_factor": 0.01, "_FillValue": -9999}
        M.Mask(ds).corr(tolerance=0.3, inplace=True)
        assert isinstance(ds["v_x"], fake_xarray.DataArray) and np.isnan(ds["v_x"].values[~m.values]).all()
        assert ds["v_x"].attrs["units"] == "m s-1" and ds["v_x"].encoding["scale_factor"] == 0.01
    finally:
        monkeypatch.undo()
        for mod in (V, F, M):
            importlib.reload(mod)


def test_pyorc_installed_branch_of_get_piv(monkeypatch):
    """The tail of the reference accessor (pyorc/api/frames.py:156-196) in ``pyorc_amd.frames.get_piv``: frames that carry a camera
    configuration (``da.frames.camera_config``) take window size and resolution from a COPY of it, and the result goes through
    ``ds.velocimetry.add_xy_coords(mesh_coords, coords, attrs)``, ``ds.attrs`` (+ the camera configuration as JSON, with an overridden
    window size) and ``ds.velocimetry.set_encoding()``.  pyorc / xarray are not installed here: doubles of exactly the members
    the branch touches (signatures as in the reference), the GPU call replaced by the oracle."""
    import importlib
    import sys
    import types

    from oracle import c_oracle
    from pyorc_amd.synth import particle_stack
    from tests import fake_xarray

    calls = {}

    class CameraConfig:
        def __init__(self):
            self.window_size, self.resolution = 25, 0.02

        def to_json(self):
            return f'{{"window_size": {self.window_size}, "resolution": {self.resolution}}}'

    class FramesAccessor:
        def __init__(self, obj, cc):
            self._obj, self.camera_config = obj, cc

        def get_piv_coords(self, window_size, search_area_size, overlap):     # pyorc/api/frames.py:60-111
            calls["get_piv_coords"] = (tuple(window_size), tuple(search_area_size), tuple(overlap))
            return {"x": None, "y": None}, {"xp": "XP", "yp": "YP", "xs": "XS", "ys": "YS", "lon": "LON", "lat": "LAT"}

    class VelocimetryAccessor:
        def __init__(self, ds):
            self.ds = ds

        def add_xy_coords(self, mesh_coords, coords, attrs):                  # pyorc/api/velocimetry.py (same argument order)
            calls["add_xy_coords"] = (mesh_coords, {k: np.asarray(v) for k, v in coords.items()}, attrs)
            return self.ds

        def set_encoding(self):
            calls["set_encoding"] = calls.get("set_encoding", 0) + 1

    monkeypatch.setattr(fake_xarray.Dataset, "velocimetry", property(lambda self: VelocimetryAccessor(self)), raising=False)
    pyorc = types.ModuleType("pyorc")
    pyorc.const = types.ModuleType("pyorc.const")
    pyorc.const.PERSPECTIVE_ATTRS, pyorc.const.GEOGRAPHICAL_ATTRS = {"xp": {"axis": "X"}}, {"lon": {"units": "degrees_east"}}
    monkeypatch.setitem(sys.modules, "pyorc", pyorc)
    monkeypatch.setitem(sys.modules, "pyorc.const", pyorc.const)
    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    import pyorc_amd.frames as F
    import pyorc_amd.velocimetry as V

    for mod in (V, F):
        importlib.reload(mod)
    try:
        monkeypatch.setattr(V.piv, "piv_pairs", __import__("tests.doubles", fromlist=["x"]).oracle_piv_pairs)
        monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
        fr = particle_stack(5, 96, 128, seed=5)
        t = np.arange(5) / 25.0
        da = fake_xarray.DataArray(fr, ("time", "y", "x"), {"time": t, "y": np.arange(96)[::-1] * 0.02, "x": np.arange(128) * 0.02}, attrs={"h_a": 1.5})
        cc = CameraConfig()
        da.frames = FramesAccessor(da, cc)
        ds = F.get_piv(da)                                     # window 25 from the camera configuration -> 24 x 24, overlap int(round(25) / 2) = 12
        assert calls["get_piv_coords"] == ((24, 24), (24, 24), (12, 12)) and calls["set_encoding"] == 1
        mesh, coords, attrs = calls["add_xy_coords"]
        assert mesh["lon"] == "LON" and attrs == {"xp": {"axis": "X"}, "lon": {"units": "degrees_east"}}
        assert coords["x"].shape == (9,) and coords["y"].shape == (7,) and ds["v_x"].values.shape == (4, 7, 9)
        assert ds.attrs["h_a"] == 1.5 and ds.attrs["camera_config"] == '{"window_size": 25, "resolution": 0.02}'
        ref = F.get_piv(fr, 25, time=t, resolution=0.02)      # the same numbers as the plain-array call with the configuration's values
        for k in ref:
            assert np.array_equal(ds[k].values, ref[k], equal_nan=True)
        ds = F.get_piv(da, window_size=32)                     # an argument overrides the COPY: the JSON carries it, the frames' own object does not
        assert calls["get_piv_coords"] == ((32, 32), (32, 32), (16, 16)) and '"window_size": 32' in ds.attrs["camera_config"]
        assert cc.window_size == 25 and calls["set_encoding"] == 2
    finally:
        monkeypatch.undo()
        for mod in (V, F):
            importlib.reload(mod)


def test_prime_factor_size_lists_agree(lib):
    """The Makefile's PFA_SIZES (one translation unit per size), common.h's LSPIV_PFA_SIZES (declarations + dispatch) and the
    instantiation files on disk name the same window sizes, and the dispatcher sends exactly those to kind 8."""
    import re

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyorc_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    sizes_mk = [int(x) for x in re.search(r"^PFA_SIZES\s*:=\s*(.*)$", mk, re.M).group(1).split()]
    hdr = open(os.path.join(csrc, "common.h")).read()
    block = re.search(r"#define LSPIV_PFA_SIZES\(X\)(.*?)\n#define", hdr, re.S).group(1)
    sizes_h = [int(x) for x in re.findall(r"X\((\d+)\)", block)]
    assert sizes_mk == sizes_h == sorted(sizes_h)
    on_disk = sorted(int(m.group(1)) for f in os.listdir(csrc) if (m := re.fullmatch(r"piv_fft(\d+)\.hip", f)))
    assert on_disk == sorted(sizes_h + [8, 16, 32, 64])
    assert sizes_h == [n for n in range(2, 65) if lib.lspiv_kernel_kind(n, n) == 8]
    assert all(n % 2 == 0 and n & (n - 1) for n in sizes_h)        # even, not a power of two


def test_walking_segments_are_anchored_to_the_absolute_pair_index(lib):
    """common.h::walk_segments: segments start at multiples of the anchor length of pair_offset + local index, so two
    chunkings with aligned boundaries run the same jobs; an off-anchor chunk gets a shorter first segment."""
    f = lib.lspiv_debug_segments

    def cut(n_pairs, offset, L):
        first, n_seg = C.c_int64(), C.c_int64()
        assert f(n_pairs, offset, L, C.byref(first), C.byref(n_seg)) == 0
        edges, p = [], 0
        for s in range(n_seg.value):
            q = min(p + (first.value if s == 0 else L), n_pairs)
            edges.append((offset + p, offset + q))
            p = q
        assert p == n_pairs
        return edges

    L = window.chunk_alignment((32, 32), (128, 160), (16, 16))      # a small grid: the window family's base anchor
    assert L == 25 == window.chunk_alignment((64, 64), (256, 256), (32, 32)) == window.chunk_alignment((24, 24), (128, 160), (12, 12))
    # without a grid (ABI 5): the alignment that is right on EVERY grid -- the longest anchor, a multiple of the short one (ABI 4: 25)
    assert window.chunk_alignment((32, 32)) == 75 == window.chunk_alignment((64, 64)) == window.chunk_alignment((24, 24)) and 75 % L == 0
    assert window.chunk_alignment((32, 16)) == 1 and window.chunk_alignment((31, 31)) == 1   # per-pair kernels
    # round 5: the anchor length depends on the window GRID -- 75 pairs where the grid has at least as many windows as the chip has
    # lane groups for that window family (6 144 at 32 x 32, 2 048 at 64 x 64, 12 288 up to 16 x 16), 25 below (75: csrc/common.h)
    ca = window.chunk_alignment
    assert ca((32, 32), (1080, 1920), (16, 16)) == 75 and ca((32, 32), (2160, 3840), (16, 16)) == 75     # 7 854 / 32 026 windows
    assert ca((32, 32), (720, 1280), (16, 16)) == 25 and ca((32, 32), (785, 875), (16, 16)) == 25         # 3 476 / 2 544
    assert ca((64, 64), (1080, 1920), (48, 48)) == 75 and ca((64, 64), (1080, 1920), (32, 32)) == 25     # 7 488 / 1 888
    assert ca((24, 24), (1080, 1920), (12, 12)) == 75 and ca((16, 16), (1080, 1920), (8, 8)) == 75      # 14 151 / 31 866
    assert ca((16, 16), (540, 960), (8, 8)) == 25 and ca((31, 31), (1080, 1920), (15, 15)) == 1           # 7 854 < 12 288; per-pair kernels
    assert window.chunk_alignment_any_grid((32, 32)) == 75 and window.chunk_alignment_any_grid((31, 31)) == 1
    assert lib.lspiv_chunk_alignment_grid(10, 10, 32, 32, 16, 16) < 0                                      # frame smaller than the window
    _lib.set_option("walk", 49)                                                                            # a forced anchor length wins everywhere
    try:
        assert ca((32, 32), (1080, 1920), (16, 16)) == 49 == ca((32, 32)) == ca((32, 32), (270, 480), (16, 16))
    finally:
        _lib.set_option("walk", 1)
    whole = cut(1000, 0, L)
    assert whole[0] == (0, 25) and whole[-1] == (975, 1000) and len(whole) == 40
    for bounds in ([0, 250, 500, 1000], [0, 25, 50, 975, 1000], [0, 100, 1000]):       # aligned chunkings: same segments
        got = [e for a, b in zip(bounds, bounds[1:]) for e in cut(b - a, a, L)]
        assert got == whole
    assert cut(40, 10, L) == [(10, 25), (25, 50)]                                        # off-anchor start: short head
    assert cut(7, 3, L) == [(3, 10)] and cut(1, 24, L) == [(24, 25)]
    assert cut(60, 0, 63) == [(0, 60)]
    assert f(0, 0, 25, None, None) == _lib.LSPIV_EINVAL


def test_aligned_chunk_slices_cover_every_pair_once():
    for n_frames, cs, align in [(1001, 334, 25), (1001, 5, 25), (21, 21, 25), (101, 50, 25), (77, 26, 1), (3, 5, 25), (52, 26, 25)]:
        sl = velocimetry.aligned_slices(n_frames, cs, align)
        assert sl[0][0] == 0 and sl[-1][1] == n_frames
        assert all(a % align == 0 and b - a >= 2 for a, b in sl)
        assert all(sl[i][1] - 1 == sl[i + 1][0] for i in range(len(sl) - 1))      # one shared (halo) frame
        assert sum(b - a - 1 for a, b in sl) == n_frames - 1
        assert all((b - a - 1) % align == 0 for a, b in sl[:-1])
        assert max(b - a - 1 for a, b in sl) <= max(align, cs)
    sl = velocimetry.aligned_slices(1001, 1001, 25, n_win=2**29)   # 32-bit window index per launch
    assert all(b - a - 1 <= 3 for a, b in sl)
    # a chunk never reads more frames than planned when the plan is at least one anchor length ...
    for cs in (26, 27, 51, 52, 334, 1001):
        assert max(b - a for a, b in velocimetry.aligned_slices(1001, cs, 25)) <= cs
    # ... and a plan below one anchor length is honoured when that anchor does not fit the memory budget (the low-memory
    # branch of the planner): chunks of chunksize - 1 pairs, off the anchors, every pair still exactly once
    sl = velocimetry.aligned_slices(101, 5, 25, fits=lambda n: False)
    assert max(b - a for a, b in sl) == 5 and sum(b - a - 1 for a, b in sl) == 100 and sl[0][0] == 0 and sl[-1][1] == 101
    assert velocimetry.aligned_slices(101, 5, 25, fits=lambda n: True) == velocimetry.aligned_slices(101, 5, 25)
    assert velocimetry.aligned_slices(101, 5, 25)[0] == (0, 26)


def test_shard_blocks_start_on_anchors():
    for n, w, al in [(8000, 8, 25), (1000, 8, 25), (1000, 3, 25), (30, 4, 25), (999, 2, 25), (100, 8, 1)]:
        blocks = [shard.pair_block(n, r, w, al) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert all(a % al == 0 for a, b in blocks if b > a)
        assert sorted(shard.block_sizes(n, w, al)) == sorted(b - a for a, b in blocks)
    assert shard.block_sizes(8000, 8, 25) == [1000] * 8


def test_streamed_chain_cuts_chunks_on_anchors(lib):
    """pipeline.CameraToVelocity._chunk_bounds: the time chunks of the streamed run start on multiples of the kernels'
    anchor length (so the chunked PIV returns the bits of one call) and cover every pair once; sizes without anchors
    (per-pair kernels) are cut evenly.  Host logic only: no device needed."""
    from types import SimpleNamespace

    from pyorc_amd.pipeline import CameraToVelocity

    for ws, n_pairs, n_chunks in (((32, 32), 200, 8), ((32, 32), 1000, 8), ((64, 64), 82, 8), ((32, 32), 24, 8), ((32, 32), 26, 3),
                                  ((128, 128), 40, 8), ((33, 33), 10, 4), ((24, 24), 999, 5)):
        b = CameraToVelocity._chunk_bounds(SimpleNamespace(window_size=ws, ortho_shape=(270, 480), overlap=(ws[0] // 2, ws[1] // 2)), n_pairs, n_chunks)
        align = lib.lspiv_chunk_alignment_grid(270, 480, ws[0], ws[1], ws[0] // 2, ws[1] // 2)   # a small grid: the window family's base anchor
        assert b[0] == 0 and b[-1] == n_pairs and all(x < y for x, y in zip(b, b[1:])), (ws, b)
        assert all(x % align == 0 for x in b[:-1]), (ws, align, b)
        assert len(b) - 1 <= max(1, n_chunks) and (len(b) - 1 == 1 or n_pairs > align)
    small = dict(ortho_shape=(270, 480), overlap=(16, 16))
    assert CameraToVelocity._chunk_bounds(SimpleNamespace(window_size=(32, 32), **small), 200, 8) == [0, 25, 50, 75, 100, 125, 150, 175, 200]
    assert CameraToVelocity._chunk_bounds(SimpleNamespace(window_size=(128, 128), ortho_shape=(540, 960), overlap=(64, 64)), 40, 8) == [0, 5, 10, 15, 20, 25, 30, 35, 40]
    # a grid with the long anchors (1080p, 7 854 windows): chunks of a multiple of 75 pairs
    assert CameraToVelocity._chunk_bounds(SimpleNamespace(window_size=(32, 32), ortho_shape=(1080, 1920), overlap=(16, 16)), 1000, 8) == [0, 150, 300, 450, 600, 750, 900, 1000]


def test_every_environment_switch_is_documented():
    """Every LSPIV_* variable the library, the Python package or bench.py reads is listed in INTEGRATION.md."""
    import glob
    import re

    names = set()
    for f in glob.glob(os.path.join(ROOT, "pyorc_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "pyorc_amd", "csrc", "*.cpp")) + \
            glob.glob(os.path.join(ROOT, "pyorc_amd", "*.py")) + \
            [os.path.join(ROOT, "bench.py")]:
        names |= set(re.findall(r'(?:getenv\(|environ\.get\(|environ\[|setdefault\()"(LSPIV_[A-Z0-9_]+)"', open(f).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 15 and not missing, missing


def test_product_has_no_torch_dependency():
    """The multi-GPU path goes through lspiv_comm_* (RCCL dlopen'ed by the C library): nothing under pyorc_amd/ or in
    bench.py imports torch (two HIP runtimes in one process was round 1's hazard)."""
    files = [os.path.join(ROOT, "bench.py")]
    for dirpath, _, names in os.walk(os.path.join(ROOT, "pyorc_amd")):
        files += [os.path.join(dirpath, f) for f in names if f.endswith(".py")]
    for f in files:
        txt = open(f).read()
        assert not re.search(r"^\s*(import torch|from torch)", txt, flags=re.M), f
    code = "import sys; import pyorc_amd, pyorc_amd.comm, pyorc_amd.shard, pyorc_amd.velocimetry; assert 'torch' not in sys.modules"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_comm_argument_errors(lib):
    h = C.c_void_p()
    buf = C.create_string_buffer(128)
    assert lib.lspiv_comm_unique_id(7, buf) == _lib.LSPIV_EINVAL
    assert lib.lspiv_comm_unique_id(1, buf) == 0 and buf.raw[:8] == b"LSPIVSHM" and any(buf.raw[8:32])
    assert lib.lspiv_comm_init(2, 2, buf, 1, C.byref(h)) == _lib.LSPIV_EINVAL
    assert lib.lspiv_comm_init(0, 1, C.create_string_buffer(128), 1, C.byref(h)) == _lib.LSPIV_EINVAL   # not an shm id
    assert lib.lspiv_comm_init(0, 1, buf, 1, C.byref(h)) == 0
    r, w, t, n = C.c_int(-1), C.c_int(-1), C.c_int(-1), C.c_int(-1)
    assert lib.lspiv_comm_info(h, C.byref(r), C.byref(w), C.byref(t), C.byref(n)) == 0
    assert (r.value, w.value, t.value, n.value) == (0, 1, 1, 1)
    a = np.arange(6, dtype=np.float32)
    out = np.empty(6, np.float32)
    assert lib.lspiv_comm_allreduce(h, _lib.ptr(a), _lib.ptr(out), 6, 1, 0) == 0 and np.array_equal(out, a)
    assert lib.lspiv_comm_allreduce(h, _lib.ptr(a), _lib.ptr(out), 6, 0, 0) == _lib.LSPIV_EINVAL     # uint8: not a collective dtype
    assert lib.lspiv_comm_allgather(h, _lib.ptr(a), _lib.ptr(out), 6, 1) == 0 and np.array_equal(out, a)
    assert lib.lspiv_comm_barrier(h) == 0
    assert lib.lspiv_comm_destroy(h) == 0


def test_bench_fails_loudly_without_a_gpu():
    """`python bench.py [--gpus N]` launches its own ranks; without a gfx950 device every rank must fail with the library's
    error (no CPU fallback, no JSON line on stdout, non-zero exit status) instead of hanging in a rendezvous."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    for extra in ([], ["--gpus", "2"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"] + extra,
                           capture_output=True, text=True, timeout=120, cwd=ROOT)
        assert r.returncode != 0 and r.stdout.strip() == ""
        assert "no gfx950" in r.stderr


def test_round_odd_option_reaches_the_python_mirror(lib):
    """A8 as a switch (round 3): `round_odd` lives in the library's option store, pyorc_amd.window.round_to_even and therefore
    frames.resolve_window follow it, the oracle has the same switch."""
    assert window.round_to_even((25, 27, 32)) == (24, 28, 32) == po.round_to_even((25, 27, 32))
    for mode in (1, 2, 0):
        _lib.set_option("round_odd", mode)
        try:
            with po.semantics(round_odd=mode):
                assert window.round_to_even((25, 27, 32)) == po.round_to_even((25, 27, 32))
                assert frames.resolve_window(25) == po.resolve_piv_args(25)
        finally:
            _lib.set_option("round_odd", 0)
    with pytest.raises(_lib.LspivError):
        _lib.set_option("round_odd", 3)
    for name, bad in (("v_sign", 2), ("norm_clip", 2), ("std_ddof", -1)):
        with pytest.raises(_lib.LspivError):
            _lib.set_option(name, bad)
    _lib.set_option("norm_clip", 0)
    try:
        assert lib.lspiv_kernel_kind(32, 32) == 3 and lib.lspiv_kernel_kind(64, 64) == 9 and lib.lspiv_chunk_alignment(32, 32) == 1
    finally:
        _lib.set_option("norm_clip", 1)
    assert lib.lspiv_kernel_kind(32, 32) == 1


def test_roofline_traffic_is_keyed_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py takes `roofline.traffic` from a committed profile summary only if that summary carries
    the hash of the kernel sources in this tree; a summary of other code (or without a hash, like round 2's) gives null."""
    import json
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = "void lspiv::piv_fft_walk_kernel<unsigned char, 32, false, false>(lspiv::PivParams)"
    launch = {"pairs": 1000, "H": 1080, "W": 1920, "window": 32, "overlap": 16}

    def write(name, **extra):
        json.dump({"tag": name, "launch": launch, "kernels": {kern: {"hbm_traffic_bytes": 2.5e9}}, **extra},
                  open(prof / f"{name}_summary.json", "w"))

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    args = ("piv_fft_walk_kernel<unsigned char, 32,", 1000, 1080, 1920, 32, 16)
    write("old")                                              # no hash: a round-2 summary
    write("other", code_hash="0123456789abcdef")              # profiled on other kernel code
    assert bench.measured_traffic(*args) is None
    write("now", code_hash=_lib.kernel_code_hash())
    got = bench.measured_traffic(*args)
    assert got == {"bytes": 2500000000, "source": "now_summary.json"}
    assert bench.measured_traffic("piv_fft_walk_kernel<unsigned char, 32,", 500, 1080, 1920, 32, 16) is None   # another launch shape
    assert len(_lib.kernel_code_hash()) == 16


def test_loaded_binary_is_tied_to_the_tree(lib, tmp_path, monkeypatch):
    """csrc/Makefile compiles the hash of every library source (and of the four kernel sources the profile
    summaries are keyed to) into the binary; _lib.load() recomputes them from the tree and refuses a binary built from other
    sources.  Here: the in-tree build matches; a copy of the sources with one edited header does not, load() raises
    LspivLibraryStale for it, and LSPIV_ALLOW_STALE=1 loads it anyway (bench.py then prints binary_hash_matches: false)."""
    import shutil

    prov = _lib.binary_provenance(lib)
    assert prov["binary_hash_matches"] and prov["binary_kernel_hash"] == _lib.kernel_code_hash() and len(prov["binary_source_hash"]) == 16
    assert prov["binary_source_hash"].encode() in lib.lspiv_version()
    assert lib.lspiv_build_info(0).decode() == prov["binary_kernel_hash"] and lib.lspiv_build_info(7) == b""
    copy = tmp_path / "csrc"
    shutil.copytree(os.path.join(ROOT, "pyorc_amd", "csrc"), copy, ignore=shutil.ignore_patterns("*.o"))
    with open(copy / "common.h", "a") as fh:
        fh.write("// edited after the build\n")
    stale = _lib.binary_provenance(lib, csrc_dir=str(copy))
    assert not stale["binary_hash_matches"]
    assert stale["tree_kernel_hash"] != prov["tree_kernel_hash"] and stale["tree_source_hash"] != prov["tree_source_hash"]
    # load() itself: same binary, edited tree
    real = _lib.source_hash
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_hash", lambda csrc_dir=None, header=None: real(str(copy)))
    monkeypatch.delenv("LSPIV_LIBRARY", raising=False)
    monkeypatch.delenv("LSPIV_ALLOW_STALE", raising=False)
    with pytest.raises(_lib.LspivLibraryStale, match="rebuild"):
        _lib.load()
    monkeypatch.setenv("LSPIV_ALLOW_STALE", "1")
    assert _lib.load().lspiv_abi_version() == 5


def test_bench_multi_gpu_contract_on_cpu(monkeypatch):
    """VERDICT r03 item 3, the part that needs no GPU: `--strong` parses (fixed 8000-pair total by default), the N > 1 step goes
    through pyorc_amd.shard (ShardedPivDev) and the self-diagnosing keys of `config.comm` are the ones the GPU plumbing test reads."""
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--strong"])
    a = bench.parse()
    assert a.strong and a.strong_pairs == 8000 and a.gpus == 8
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    assert not bench.parse().strong
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "shard.ShardedPivDev(" in src and "comm.allgather_dev(outs" not in src      # no second, bench-only exchange path
    for key in ("kernel_ms_while_gather_in_flight", "gather_ms_overlapped", "exposed_comm_ms", "allgather_ms_alone", "rccl_env",
                "allgather_bytes_received_per_rank_per_step", "survey_8e_bytes_per_rank"):
        assert f'"{key}"' in src, key
    # the plan's cut: weak = blocks of --pairs, strong = the walking kernels' anchors
    assert shard.block_sizes(8000, 8, 25) == [1000] * 8 and shard.block_sizes(120, 2, 60) == [60, 60]
    assert shard.block_sizes(150, 2, 25) == [75, 75]


def test_float64_staging_conversion_and_offset_rule(lib):
    """Host side of VERDICT r03 item 8 (csrc/host_stage.cpp, no GPU needed): the staging threads narrow float64 frames exactly like
    numpy's astype(float32) (IEEE round to nearest; AVX2 path, ragged sizes, several frames per batch); a frame riding on a large DC
    offset has an integer estimate of it removed first -- a function of the frame alone --, small offsets are left alone, so 8-bit-like
    stacks stay bit-identical to their float32 copies."""
    import time

    rng = np.random.default_rng(8)
    for n_frames, elems in ((1, 7), (3, 1025), (2, 640 * 480 + 13), (5, 1 << 20)):
        fr = rng.normal(30.0, 60.0, (n_frames, elems)) * (1.0 + 1e-9)
        out = np.empty((n_frames, elems), np.float32)
        off = np.empty(n_frames, np.float64)
        nt = lib.lspiv_debug_narrow(_lib.ptr(fr), elems, n_frames, 1024, _lib.ptr(out), _lib.ptr(off))
        assert nt >= 1 and np.all(off == 0.0) and np.array_equal(out, fr.astype(np.float32))
    fr = rng.normal(0.0, 1.0, (4, 300 * 200)) + np.array([1.0e4, -7.3e5, 900.0, 2.0e9])[:, None]
    out = np.empty(fr.shape, np.float32)
    off = np.empty(4, np.float64)
    lib.lspiv_debug_narrow(_lib.ptr(fr), fr.shape[1], 4, 1024, _lib.ptr(out), _lib.ptr(off))
    assert off[2] == 0.0 and np.all(off[[0, 1, 3]] == np.rint(off[[0, 1, 3]])) and np.all(np.abs(off[[0, 1, 3]] - fr[[0, 1, 3]].mean(axis=1)) < 1.0)
    assert np.array_equal(out, (fr - off[:, None]).astype(np.float32))
    # what the rule is for: the texture survives the conversion (plain narrowing maps all of 2e9 + noise onto 2e9)
    assert np.abs(out[3].astype(np.float64) - (fr[3] - off[3])).max() < 1e-6 and np.abs(fr[3].astype(np.float32).astype(np.float64) - fr[3]).max() > 1.0
    again = np.empty_like(out)
    lib.lspiv_debug_narrow(_lib.ptr(fr[:2]), fr.shape[1], 2, 1024, _lib.ptr(again[:2]), None)     # a frame's offset does not depend on its chunk
    assert np.array_equal(again[:2], out[:2])
    lib.lspiv_debug_narrow(_lib.ptr(fr), fr.shape[1], 4, -1, _lib.ptr(again), _lib.ptr(off))       # switched off
    assert np.all(off == 0.0) and np.array_equal(again, fr.astype(np.float32))
    big = rng.random((8, 1080 * 1920))
    dst = np.empty(big.shape, np.float32)
    lib.lspiv_debug_narrow(_lib.ptr(big), big.shape[1], 8, 1024, _lib.ptr(dst), None)
    t0 = time.perf_counter()
    lib.lspiv_debug_narrow(_lib.ptr(big), big.shape[1], 8, 1024, _lib.ptr(dst), None)
    dt = time.perf_counter() - t0
    assert np.array_equal(dst, big.astype(np.float32))
    print(f"narrowing 8 x 1080p float64 frames: {big.nbytes / dt / 1e9:.1f} GB/s read on {nt} threads")


def test_a_binary_older_than_the_header_is_reported_as_stale(tmp_path, monkeypatch):
    """a build that lacks entry points the header declares must say "rebuild" (LspivLibraryStale), not AttributeError --
    also when its hashes cannot tell (here: a stub that reports the TREE's hashes but exports nothing else)."""
    import subprocess

    src = tmp_path / "stub.c"
    src.write_text('const char* lspiv_build_info(int k) { return k == 0 ? "%s" : k == 1 ? "%s" : ""; }\n'
                   % (_lib.kernel_code_hash(), _lib.source_hash()))
    so = tmp_path / "libstub.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    monkeypatch.delenv("LSPIV_LIBRARY", raising=False)
    monkeypatch.delenv("LSPIV_ALLOW_STALE", raising=False)
    with pytest.raises(_lib.LspivLibraryStale, match="does not export lspiv_"):
        _lib.load()
    # and one whose hashes differ is caught by them first, before any symbol is looked up
    src.write_text('const char* lspiv_build_info(int k) { return k < 2 ? "0123456789abcdef" : ""; }\n')
    so = tmp_path / "libstub2.so"                      # (another path: the loader caches a loaded object by name)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    with pytest.raises(_lib.LspivLibraryStale, match="was built from sources with hash 0123456789abcdef"):
        _lib.load()


def test_locks_are_per_device(lib):
    """the host-entry-point, launch and list locks are per device, so one process can drive several GPUs from
    several threads.  The test hook holds a lock without touching HIP: two threads holding the SAME lock of two devices overlap,
    of one device they queue; different locks of one device are independent of each other."""
    import threading

    def held(pairs, ms=150):
        ts = [threading.Thread(target=lambda d=d, w=w: lib.lspiv_debug_hold_lock(d, w, ms)) for d, w in pairs]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    import time
    for which in (0, 1, 2):
        same = held([(0, which), (0, which)])
        other = held([(0, which), (1, which)])
        assert same >= 0.29, (which, same)
        assert other <= 0.25, (which, other)
    assert held([(3, 0), (3, 1), (3, 2)]) <= 0.25
    assert held([(d, 0) for d in range(8)]) <= 0.3          # eight ranks' worth of host threads in one process
    assert lib.lspiv_debug_hold_lock(64, 0, 1) == _lib.LSPIV_EINVAL and lib.lspiv_debug_hold_lock(0, 5, 1) == _lib.LSPIV_EINVAL
    # round 6: the host-pointer projection entry points have two slots of their own (locks 3 and 4): a projection block queues behind
    # neither a PIV host call (lock 0) nor the other slot
    assert held([(0, 0), (0, 3), (0, 4)]) <= 0.25 and held([(0, 3), (0, 3)]) >= 0.29


def test_rccl_channel_cap_is_scoped_to_communicator_creation(monkeypatch):
    """NCCL_MAX_NCHANNELS is set only around ncclCommInitRank and the environment is put back."""
    from pyorc_amd import comm

    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("LSPIV_RCCL_MAX_NCHANNELS", raising=False)
    with comm._rccl_channel_cap(True):
        assert os.environ["NCCL_MAX_NCHANNELS"] == "16"
    assert "NCCL_MAX_NCHANNELS" not in os.environ and comm._rccl_channel_cap.applied == "16"
    monkeypatch.setenv("LSPIV_RCCL_MAX_NCHANNELS", "8")
    with comm._rccl_channel_cap(True):
        assert os.environ["NCCL_MAX_NCHANNELS"] == "8"
    assert "NCCL_MAX_NCHANNELS" not in os.environ
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "32")             # the user's own setting is neither overridden nor removed
    with comm._rccl_channel_cap(True):
        assert os.environ["NCCL_MAX_NCHANNELS"] == "32"
    assert os.environ["NCCL_MAX_NCHANNELS"] == "32" and comm._rccl_channel_cap.applied == "32"
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    with comm._rccl_channel_cap(False):                        # other transports: untouched
        assert "NCCL_MAX_NCHANNELS" not in os.environ


def test_bench_reads_clock_and_power_from_rocm_smi_text(monkeypatch):
    """bench.py's sampler of the sustained loop parses `rocm-smi --showpower --showclocks` (text as the tool prints it on the MI355X boxes);
    a missing tool or another format gives no samples and None figures, never an exception."""
    import subprocess as sp
    import time as _t

    import bench

    text = ("GPU[0]\t\t: fclk clock level: 0: (1400Mhz)\nGPU[0]\t\t: mclk clock level: 3: (2000Mhz)\nGPU[0]\t\t: sclk clock level: 1: (2293Mhz)\n"
            "GPU[0]\t\t: socclk clock level: 3: (1143Mhz)\nGPU[0]\t\t: Current Socket Graphics Package Power (W): 1334.0\n")

    class R:
        stdout = text

    monkeypatch.setattr(sp, "run", lambda *a, **k: R())
    t0 = _t.time() - 2.0
    with bench.SmiSampler(period=0.05) as smi:
        _t.sleep(0.3)
    s = smi.summary(t0, _t.time() + 1)
    assert s["samples"] >= 2 and s["sclk_mhz_mean"] == 2293.0 and s["socket_power_w_mean"] == 1334.0 and s["sclk_mhz_min"] == 2293
    monkeypatch.setattr(sp, "run", lambda *a, **k: (_ for _ in ()).throw(FileNotFoundError("rocm-smi")))
    with bench.SmiSampler(period=0.05) as smi2:
        _t.sleep(0.15)
    assert smi2.summary(t0, _t.time() + 1) == {"samples": 0, "sclk_mhz_mean": None, "socket_power_w_mean": None}


def test_sharded_piv_cuts_rank_blocks_on_the_grids_anchor(lib, monkeypatch):
    """shard.sharded_piv picks its default alignment from the window grid when it is told the frame shape (25 pairs on small grids,
    75 where the walking kernels use long anchors) and falls back to the window family's longest anchor -- right for every grid --
    when it is not; an explicit `align` wins.  One rank, the oracle as compute: host logic only."""
    from pyorc_amd import shard
    from pyorc_amd.synth import particle_stack
    from tests.doubles import oracle_piv_pairs

    seen = []
    real = shard.frame_block
    monkeypatch.setattr(shard, "frame_block", lambda n, r, w, align: (seen.append(align), real(n, r, w, align))[1])

    class OneRank:
        rank, world = 0, 1

        def allreduce(self, a, op=None):
            return a

        def allgather(self, a):
            return np.asarray(a)[None]

    stack = particle_stack(6, 96, 128, seed=1)
    compute = lambda fr, ws, ov, thr, pair_offset=0: oracle_piv_pairs(fr, ws, ov, thr)
    ref = np.stack(oracle_piv_pairs(stack, (32, 32), (16, 16)))
    for kw, want in ((dict(), 75), (dict(frame_shape=(96, 128)), 25), (dict(frame_shape=(1080, 1920)), 75), (dict(align=7), 7),
                     (dict(frame_shape=(1080, 1920), align=25), 25)):
        full = shard.sharded_piv(lambda a, b: stack[a:b], 5, (32, 32), (16, 16), OneRank(), compute=compute, **kw)
        assert seen[-1] == want, (kw, seen[-1])
        assert np.array_equal(full, ref, equal_nan=True)
    assert shard.block_sizes(8000, 8, 75) == [975, 975, 1050, 975, 975, 1050, 975, 1025] and sum(shard.block_sizes(8000, 8, 75)) == 8000
