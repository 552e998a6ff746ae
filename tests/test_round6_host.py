"""Round 6, host side (CPU): the lazy-aware planner, the adaptive prefetch depth, the device the worker threads select, the hand-off
from ``project_hip`` to ``get_ffpiv`` (doubles of xarray + dask: tests/lazy_doubles.py; the real dask: tests/test_real_dask.py), and the
deferred ``install()``."""
import importlib
import os
import subprocess
import sys
import textwrap
import threading
import time
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyorc_amd import executor, resident  # noqa: E402


class ShapeOnly:
    """A lazy stack nobody loads: what the planner looks at."""

    def __init__(self, shape, dtype, blocks=None):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        if blocks:
            self.chunks = (tuple(blocks),) + tuple((n,) for n in shape[1:])

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, k):
        return np.empty(self.shape[1:], self.dtype) if not isinstance(k, slice) else self

    def load(self):
        raise AssertionError("the planner must not load anything")


def _plan(frames, ws=(32, 32), ov=(16, 16), chunksize=None, prefetch=None, host=64e9, hbm=250e9, memory_factor=4):
    from pyorc_amd import velocimetry as V, window

    dim = frames.shape[1:]
    n_rows, n_cols = window.get_array_shape(dim, ws, ov)
    return V.plan_lazy(frames, len(frames), dim, ws, ov, n_rows * n_cols, chunksize, memory_factor, "hip", prefetch,
                       host_available=host, hbm_available=hbm)


def test_a_lazy_1080p_float64_stack_is_planned_against_host_memory(lib):
    """1 000 lazy float64 1080p frames (16.6 MB each: 16.6 GB, as project_numpy hands them over) on a 64 GB host:
    several loads, on dask's 20-frame block boundaries, and what the executor may hold at its deepest -- max_depth + 1 loads -- stays
    inside available / memory_factor, the reference's budget (ffpiv.py:129)."""
    fr = ShapeOnly((1000, 1080, 1920), np.float64, blocks=[20] * 50)
    plan = _plan(fr)
    loads = plan["loads"][0]
    assert plan["windows"] == [(0, 1000)] and len(loads) >= 4                   # everything resident in HBM (8.3 GB narrowed), several loads
    assert loads[0][0] == 0 and loads[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(loads, loads[1:]))     # no halo: every frame once
    assert all(f0 % 20 == 0 for f0, _ in loads)                                  # cut where dask cuts
    assert plan["depth"] is None and plan["max_depth"] == executor.max_depth() == 4
    frame = 1080 * 1920 * 8
    assert plan["host_frame_bytes"] == frame and plan["peak_host_bytes"] == 5 * max(b - a for a, b in loads) * frame
    assert plan["peak_host_bytes"] <= plan["host_budget"] == 16e9
    assert plan["align"] == 75 and plan["load_frames"] == 60 and len(loads) == 17  # 999 pairs = 14 anchors -> granule one anchor = 75 (host: 192) -> whole blocks
    # the same stack with the depth fixed by the caller: (depth + 1) loads share the budget
    p1 = _plan(fr, prefetch=1)
    assert p1["depth"] == 1 and p1["max_depth"] == 1 and p1["load_frames"] == 60 and p1["peak_host_bytes"] == 2 * 60 * frame <= p1["host_budget"]
    # a host with a quarter of that memory: the budget binds (48 frames a load at the adaptive depth), still whole blocks
    small = _plan(fr, host=16e9)
    assert small["load_frames"] == 40 and small["peak_host_bytes"] <= small["host_budget"] == 4e9
    # a host that is nearly full: the reference's warning (its text, ffpiv.py:131-135), and 5 frames per load it is
    with pytest.warns(UserWarning, match=r"Memory availability is poor \(0\.2 GB\)\. Chunk size is automatically set to 2 to avoid"):
        poor = _plan(fr, host=0.8e9)
    assert poor["load_frames"] == 5 and max(b - a for a, b in poor["loads"][0]) == 5
    # a user's chunksize is the load size; below 2 the reference's OverflowError
    assert _plan(fr, chunksize=50)["load_frames"] == 50 and max(b - a for a, b in _plan(fr, chunksize=50)["loads"][0]) == 40    # whole blocks
    with pytest.raises(OverflowError, match="Chunk size with selected nr of chunks"):
        _plan(fr, chunksize=1)
    # a stack beyond the HBM budget: windows cut on lcm(anchor, block) = 300 pairs, consecutive windows share their halo frame
    tight = _plan(fr, hbm=4 * 3.5e9)
    assert tight["windows"] == [(0, 301), (300, 601), (600, 901), (900, 1000)]
    assert all(ls[0][0] == w0 and ls[-1][1] == w1 for ls, (w0, w1) in zip(tight["loads"], tight["windows"]))
    # a short stack: fewer anchors than MIN_LOADS -> one anchor per load, still no single-chunk plan
    few = _plan(ShapeOnly((201, 1080, 1920), np.float64))
    assert [b - a for a, b in few["loads"][0]] == [67, 67, 67] and few["align"] == 75


def test_plan_loads_covers_every_frame_once_for_any_blocks():
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(2, 400))
        L = int(rng.integers(1, 90))
        first = int(rng.integers(0, max(1, n - 1)))
        blocks = None
        if rng.random() < 0.7:
            lens = []
            while sum(lens) < n:
                lens.append(int(rng.integers(1, 40)))
            lens[-1] -= sum(lens) - n
            blocks = [0] + list(np.cumsum([x for x in lens if x > 0]))
        loads = resident.plan_loads(n, L, blocks, first=first)
        assert loads[0][0] == first and loads[-1][1] == n and all(a[1] == b[0] for a, b in zip(loads, loads[1:]))
        assert all(0 < b - a <= L for a, b in loads)
        if blocks:   # a cut inside a block only where the block itself is longer than a load
            for a, b in loads[:-1]:
                if b not in blocks:
                    nxt = min(x for x in blocks if x > a)
                    assert nxt - a > L, (loads, blocks, L)


def test_time_blocks_reads_xarray_and_dask_chunk_tuples():
    class A:
        def __init__(self, chunks, n):
            self.chunks, self._n = chunks, n

        def __len__(self):
            return self._n

    assert resident.time_blocks(A(((20, 20, 5), (8,), (9,)), 45)) == [0, 20, 40, 45]
    assert resident.time_blocks(A(None, 45)) is None and resident.time_blocks(A(((20, 20), (8,)), 45)) is None     # does not cover the axis
    assert resident.time_blocks(np.zeros((4, 2, 2))) is None


def test_depth_adapts_to_a_single_threaded_loader_and_backs_off_a_parallel_one():
    """The depth follows the run: loads that take 3 x a launch and scale with threads -> three side by side, the consumer stops
    waiting; loads that do NOT scale (one lock inside: a loader that is already parallel, or a single decoder) -> one step up, no
    gain measured, back and frozen.  Results in order either way."""
    def run(load, n=14, consume=0.02, **kw):
        pf = executor.ChunkPrefetcher(list(range(n)), load, **kw)
        got = []
        t0 = time.perf_counter()
        for k, v in pf:
            got.append(v)
            time.sleep(consume)
        return pf, got, time.perf_counter() - t0

    def scalable(k):
        time.sleep(0.06)
        return k * k

    pf, got, wall = run(scalable)
    assert got == [k * k for k in range(14)] and pf.adaptive and pf.workers == pf.max_depth == 4
    assert pf.depth_history[0] == 1 and pf.depth_history[1] in (3, 4) and max(pf.depth_history) <= 4, pf.depth_history     # ceil(0.06 / 0.02), give or take the sleeps' overshoot
    assert wall < 0.06 + 14 * 0.02 + 0.25 and wall < 0.7 * 14 * 0.08, wall            # ~ launches alone; the serial loop: 14 x 0.08
    gate = threading.Lock()

    def serial_inside(k):
        with gate:
            time.sleep(0.03)
        return -k

    pf, got, wall = run(serial_inside, n=20, consume=0.01)
    assert got == [-k for k in range(20)]
    assert pf._frozen and pf.depth_history[-1] < max(pf.depth_history), pf.depth_history
    # a fixed depth is a fixed depth
    pf, _, _ = run(scalable, n=5, depth=1)
    assert not pf.adaptive and set(pf.depth_history) == {1} and pf.workers == 1
    pf, _, _ = run(scalable, n=3, depth=0)
    assert set(pf.depth_history) == {0}


def test_worker_threads_and_projection_blocks_select_the_callers_device(monkeypatch):
    """hipSetDevice is per thread (ADVICE r05 / VERDICT item 2): the loader threads take the device of the thread that created the
    prefetcher, a project_hip block the device its graph was built on -- before anything else they do."""
    from pyorc_amd import plugin

    seen = []
    monkeypatch.setattr(executor, "current_device", lambda: 3)
    monkeypatch.setattr(executor, "bind_device", lambda d: seen.append((threading.current_thread().name, d)))

    def load(k):
        assert any(n == threading.current_thread().name for n, _ in seen), "load ran before the thread selected its device"
        return k

    pf = executor.ChunkPrefetcher(list(range(6)), load, depth=2, workers=2)
    assert pf.device == 3 and [v for _, v in pf] == list(range(6))
    assert seen and all(d == 3 and n.startswith("lspiv-load") for n, d in seen) and len(seen) <= 2      # once per worker thread
    assert executor.ChunkPrefetcher([], load, device=None).device is None

    class Plan:
        def project_frames(self, a, keep_uint8=None):
            assert seen[-1] == (threading.current_thread().name, 5)      # the block selected its device before it touched the plan
            return np.zeros((a.shape[0], 4, 6), np.float32)

    seen.clear()
    monkeypatch.setattr(plugin, "_projection_plan", lambda src, dst, args, device=None: Plan())
    out = plugin._project_block(np.zeros((2, 8, 8), np.uint8), plan_args=(None,) * 5, dst_shape=(4, 6), device=5)
    assert out.shape == (2, 4, 6) and seen == [(threading.current_thread().name, 5)]


class OraclePlan:
    """pyorc_amd.project.Projection computed by the numpy oracle; ``project_into`` on the HostStack double."""

    made = []

    def __init__(self, src_shape, dst_shape, *m):
        OraclePlan.made.append(self)
        self.src_shape, self.dst_shape, self.m, self.blocks, self.into = tuple(src_shape), tuple(dst_shape), m, [], []

    def _go(self, frames):
        from oracle import project_oracle as pro

        return pro.project_frames(np.asarray(frames), self.dst_shape, *self.m).astype(np.float32)

    def project_frames(self, frames, keep_uint8=None):
        self.blocks.append(frames.shape[0])
        return self._go(frames)

    def project_into(self, frames, out, f0=0):
        self.into.append((int(f0), len(frames)))
        out.arr[f0:f0 + len(frames)] = self._go(np.asarray(frames))

    def close(self):
        pass


def test_project_hip_hands_the_camera_frames_to_get_ffpiv(monkeypatch):
    """frames.project(method="hip") followed by get_piv(engine="hip") -- with only Frames.project's own fillna(0.0)
    in between -- loads the CAMERA blocks (each exactly once), projects them into the resident stack and never computes a projected
    block on the host; one more layer in between (anything) and the generic path runs: every projected block once, no halo block
    twice.  Same bits both ways."""
    from pyorc_amd import _lib, frames as F, plugin, project as P, velocimetry as V
    from pyorc_amd.synth import particle_stack, projection_maps
    from tests import doubles, lazy_doubles

    monkeypatch.setitem(sys.modules, "xarray", lazy_doubles)
    monkeypatch.setattr(P, "Projection", OraclePlan)
    monkeypatch.setattr(_lib, "require_device", lambda: None)
    monkeypatch.setattr(V.piv, "piv_pairs", doubles.oracle_piv_pairs)
    monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
    monkeypatch.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 10)
    stacks = doubles.use_host_stacks(monkeypatch)
    OraclePlan.made = []
    plugin.uninstall()
    src, dst = (96, 128), (72, 100)
    maps = projection_maps(src, dst, tilt=0.2, seed=4)
    cam = particle_stack(47, src[0], src[1], seed=12)
    t = np.arange(47) / 30.0
    try:
        video = lazy_doubles.from_frames(cam, block=10)
        counted = video.map_time(lambda blk: blk, "normalize")            # a per-block stage before the projection: counts camera blocks
        ortho = lazy_doubles.frames_project(counted, maps, dst, plugin.project_hip)
        assert ortho.dtype == np.float32 and ortho.shape == (47,) + dst and ortho.data.name.startswith("where-")
        hit = plugin.hip_projection_source(ortho)
        assert hit is not None and hit["source"] is counted and hit["dst_shape"] == dst
        got = F.get_piv(ortho, 32, time=t, resolution=0.01)
        st = executor.LAST_STATS
        assert st["plan"]["source"] == "camera" and st["chunks"] == 5 and st["plan"]["load_frames"] == 10     # 46 pairs, anchors of 10: loads of one whole block each
        per_layer = lambda prefix: {k[1]: v for k, v in counted.calls.items() if k[0].startswith(prefix)}    # noqa: E731 (block -> computations)
        assert per_layer("normalize") == {i: 1 for i in range(5)} and per_layer("project_block") == {}      # camera blocks once; no projected block at all
        plan = OraclePlan.made[0]
        assert len(OraclePlan.made) == 1 and plan.blocks == [] and sorted(plan.into) == [(0, 10), (10, 10), (20, 10), (30, 10), (40, 7)]      # no projected block on the host
        assert stacks.uploads == []                                             # the ortho frames were never uploaded: they were made in place
        # anything else in between -> the generic path: the projected frames are computed by the blocks and uploaded
        other = ortho.map_time(lambda blk: blk, "astype")
        assert plugin.hip_projection_source(other) is None and plugin.hip_projection_source(ortho[0:20]) is None
        counted.calls.clear(); plan.blocks.clear(); plan.into.clear()
        ref = F.get_piv(other, 32, time=t, resolution=0.01)
        assert executor.LAST_STATS["plan"]["source"] == "frames" and plan.into == [] and sorted(plan.blocks) == [7, 10, 10, 10, 10]
        assert per_layer("normalize") == per_layer("project_block") == {i: 1 for i in range(5)}     # every block decoded + projected ONCE: no halo block twice
        assert sorted(stacks.uploads) == [(0, 10), (10, 10), (20, 10), (30, 10), (40, 7)]
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), k
        # ... and both equal the materialised stack's result (the reference's independent windows)
        whole = F.get_piv(OraclePlan(src, dst, *maps)._go(cam), 32, time=t, resolution=0.01)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(got[k], whole[k], equal_nan=True), k
        # ensemble mode takes the same road
        from tests.test_shard_gloo import OracleEnsemble

        class Ens(OracleEnsemble):
            def accumulate(self, frames, corr_min, s2n_min, thr=None, out=None):
                return super().accumulate(np.asarray(frames), corr_min, s2n_min, thr, out)

            def finish(self, count_min, n_frames):
                mean = self.s / np.maximum(self.k, 1)[:, None, None]
                u, v = self.po.u_v_displacement(mean[None], self.n_rows, self.n_cols)
                return u.astype(np.float32), v.astype(np.float32), self.k.astype(np.float32)

            def close(self):
                pass

        monkeypatch.setattr(V.piv, "Ensemble", Ens)
        plan.into.clear()
        e1 = F.get_piv(ortho, 32, time=t, resolution=0.01, ensemble_corr=True)
        assert sorted(plan.into) == [(0, 10), (10, 10), (20, 10), (30, 10), (40, 7)]
        e2 = F.get_piv(other, 32, time=t, resolution=0.01, ensemble_corr=True)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(e1[k], e2[k], equal_nan=True), k
    finally:
        plugin.uninstall()


def test_a_graph_that_is_not_fillna_of_project_hip_is_not_short_cut():
    """The matcher accepts the registered node itself and xarray's fillna pattern on it -- nothing else."""
    from pyorc_amd import plugin
    from tests.lazy_doubles import ArrayHandle, Graph

    class Obj:
        def __init__(self, name, deps, shape=(10, 4, 6), dtype=np.float32):
            self.data, self.shape, self.dtype = ArrayHandle(name, None, Graph(deps)), shape, np.dtype(dtype)

        def __len__(self):
            return self.shape[0]

    src = Obj("video-1", {}, (10, 8, 8), np.uint8)
    plugin._PROJECTIONS.clear()
    try:
        plugin._register_projection(Obj("project_block-aa", {}), src, (None,) * 5, (4, 6), None)
        P = "project_block-aa"
        ok = {"where-1": {"invert-1", P}, "invert-1": {"isnan-1"}, "isnan-1": {P}, P: {"video-1"}}
        assert plugin.hip_projection_source(Obj(P, {P: {"video-1"}}))["source"] is src
        assert plugin.hip_projection_source(Obj("where-1", ok))["source"] is src
        assert plugin.hip_projection_source(Obj("where-1", {**ok, "invert-1": {"notnull-9"}, "notnull-9": {P}})) is None
        assert plugin.hip_projection_source(Obj("where-2", {"where-2": {"notnull-2", P}, "notnull-2": {P}}))["source"] is src
        for bad in ({"where-1": {"invert-1", "other-3"}, "invert-1": {"isnan-1"}, "isnan-1": {P}},         # the data operand is something else
                    {"where-1": {"invert-1", P}, "invert-1": {"isnan-1"}, "isnan-1": {"other-3"}},          # the mask comes from elsewhere
                    {"where-1": {"invert-1", P, "x-1"}, "invert-1": {"isnan-1"}, "isnan-1": {P}},           # a third operand
                    {"where-1": {"gt-1", P}, "gt-1": {P}}):                                                  # not a NaN mask
            assert plugin.hip_projection_source(Obj("where-1", bad)) is None, bad
        assert plugin.hip_projection_source(Obj("add-1", {"add-1": {P}})) is None
        assert plugin.hip_projection_source(Obj("where-1", ok, dtype=np.float64)) is None                 # a dtype change after project
        assert plugin.hip_projection_source(Obj("where-1", ok, shape=(9, 4, 6))) is None                   # a time selection
        assert plugin.hip_projection_source(np.zeros((3, 4, 6), np.float32)) is None
    finally:
        plugin._PROJECTIONS.clear()


def test_import_pyorc_amd_does_not_import_pyorc_and_installs_when_pyorc_is_imported(tmp_path):
    """``import pyorc_amd`` used to import pyorc (xarray, dask, cv2, numba ...) in every process and to swallow a failing
    installation.  Now: pyorc already imported -> patched at once; importable -> patched right after its own import (a post-import
    hook), not before; a pyorc that breaks the installation -> a RuntimeWarning that says so."""
    pkg = tmp_path / "pyorc"
    (pkg / "api").mkdir(parents=True)
    (pkg / "velocimetry").mkdir()
    (pkg / "__init__.py").write_text("from . import api, velocimetry, project\nMARK = 'real package body ran'\n")
    (pkg / "project.py").write_text("def project_numpy(da, cc, x, y, z, reducer='mean'):\n    return 'numpy'\n")
    (pkg / "api" / "__init__.py").write_text("from .frames import Frames\n")
    (pkg / "api" / "frames.py").write_text(textwrap.dedent("""
        from pyorc.velocimetry import ffpiv
        class Frames:
            def get_piv(self, window_size=None, overlap=None, engine="numba", ensemble_corr=False, **kwargs):
                if engine not in ["numba", "numpy"]:
                    raise ValueError(f"Selected PIV engine {engine} does not exist.")
                return ffpiv.get_ffpiv(None, None, None, None, engine=engine, **kwargs)
    """))
    (pkg / "velocimetry" / "__init__.py").write_text("from .ffpiv import get_ffpiv\n")
    (pkg / "velocimetry" / "ffpiv.py").write_text("def get_ffpiv(frames, y, x, dt, engine='numba', **kw):\n    return ('cpu', engine)\n")
    code = textwrap.dedent(f"""
        import sys, warnings
        sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {ROOT!r})
        import pyorc_amd
        from pyorc_amd import plugin
        assert "pyorc" not in sys.modules and not plugin.is_installed() and plugin._hook is not None, "import pyorc_amd imported pyorc"
        import pyorc
        assert pyorc.MARK == 'real package body ran' and plugin.is_installed() and plugin._hook is None
        assert pyorc.project.project_hip is plugin.project_hip and hasattr(pyorc.api.frames.Frames.get_piv, "__lspiv_original__")
        assert pyorc.api.frames.Frames().get_piv(engine="numpy") == ('cpu', 'numpy')
        assert pyorc.velocimetry.get_ffpiv is pyorc.velocimetry.ffpiv.get_ffpiv and hasattr(pyorc.velocimetry.get_ffpiv, "__lspiv_original__")
        plugin.uninstall()
        assert plugin.auto_install() is True and plugin.is_installed()          # pyorc is imported now: at once
        plugin.uninstall()
        # a pyorc the patch does not fit: a warning, not silence
        del pyorc.api.frames.Frames.get_piv
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert plugin.auto_install() is False
        assert any("could not register engine='hip'" in str(x.message) for x in w), [str(x.message) for x in w]
        print("OK")
    """)
    env = {k: v for k, v in os.environ.items() if k != "LSPIV_NO_AUTO_INSTALL"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_get_piv_wrapper_binds_engine_by_the_originals_signature():
    """the wrapper no longer hard-codes (window_size, overlap, engine, ensemble_corr): `engine` is found wherever the
    installed pyorc's own signature has it, positional or keyword, and everything else passes through untouched."""
    from pyorc_amd import _lib, plugin

    calls = []

    class Frames:
        def get_piv(self, window_size=None, engine="numba", overlap=None, *, extra=1, **kwargs):    # another release: another order
            calls.append((window_size, engine, overlap, extra, kwargs, plugin._route_hip.get()))
            return "done"

    wrapped = plugin._wrap_get_piv(Frames.get_piv)
    f = Frames()
    assert wrapped(f, 64, "numpy", extra=2) == "done" and calls[-1] == (64, "numpy", None, 2, {}, False)
    assert wrapped(f) == "done" and calls[-1][1] == "numba" and calls[-1][5] is False
    saved = _lib.require_device
    _lib.require_device = lambda: None
    try:
        assert wrapped(f, 64, "hip", (8, 8), chunksize=5) == "done"
        assert calls[-1] == (64, "numba", (8, 8), 1, {"chunksize": 5}, True)              # the gate sees an engine it knows; the route is set
        assert wrapped(f, engine="hip", window_size=32) == "done" and calls[-1][:2] == (32, "numba") and calls[-1][5] is True
    finally:
        _lib.require_device = saved
    assert plugin._route_hip.get() is False
    with pytest.raises(TypeError):
        wrapped(f, 1, 2, 3, 4, 5)


@pytest.mark.parametrize("ensemble", [False, True])
def test_a_lazy_stack_beyond_the_hbm_budget_runs_in_windows_with_the_same_result(monkeypatch, ensemble):
    """A lazy stack that does not fit `lspiv_available_bytes / memory_factor` is cut into resident windows on the anchors, each
    re-loading ONE halo frame; pairs, order, time stamps and bits are those of the one-window run and of the materialised stack."""
    from pyorc_amd import frames as F, velocimetry as V
    from pyorc_amd.synth import particle_stack
    from tests import doubles
    from tests.test_shard_gloo import OracleEnsemble

    class Ens(OracleEnsemble):
        def accumulate(self, frames, corr_min, s2n_min, thr=None, out=None):
            return super().accumulate(np.asarray(frames), corr_min, s2n_min, thr, out)

        def finish(self, count_min, n_frames):
            mean = self.s / np.maximum(self.k, 1)[:, None, None]
            u, v = self.po.u_v_displacement(mean[None], self.n_rows, self.n_cols)
            return u.astype(np.float32), v.astype(np.float32), self.k.astype(np.float32)

        def close(self):
            pass

    log = []

    class Lazy:
        def __init__(self, data, lo=0):
            self._d, self.lo, self.dtype, self.shape = data, lo, data.dtype, data.shape

        def __len__(self):
            return len(self._d)

        def __getitem__(self, key):
            if isinstance(key, slice):
                a, b, _ = key.indices(len(self._d))
                return Lazy(self._d[key], self.lo + a)
            return self._d[key]

        def load(self):
            log.append((self.lo, self.lo + len(self._d)))
            return np.array(self._d)

    monkeypatch.setattr(V.piv, "piv_pairs", doubles.oracle_piv_pairs)
    monkeypatch.setattr(V.piv, "Ensemble", Ens)
    monkeypatch.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 5)
    doubles.use_host_stacks(monkeypatch)
    fr = particle_stack(33, 64, 96, seed=4).astype(np.float32)
    t = np.arange(33) / 25.0
    kw = dict(time=t, resolution=0.02, ensemble_corr=ensemble)
    monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
    ref = F.get_piv(fr, 32, **kw)
    one = F.get_piv(Lazy(fr), 32, **kw)
    assert executor.LAST_STATS["plan"]["windows"] == [(0, 33)]
    # room for 12 float32 frames + their results: windows of 10 pairs (two anchors), 11 frames each
    per_frame = 64 * 96 * 4 + 16 * 3 * 5
    monkeypatch.setattr(V.window, "available_memory", lambda: 4 * (12 * per_frame + 64))
    log.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")        # (the materialised planner's low-memory warning: it plans with the same figure)
        got = F.get_piv(Lazy(fr), 32, **kw)
    assert executor.LAST_STATS["plan"]["windows"] == [(0, 11), (10, 21), (20, 31), (30, 33)]
    loaded = sorted(log)
    assert loaded[0][0] == 0 and loaded[-1][1] == 33
    assert sum(b - a for a, b in loaded) == 33 + 3                       # every frame once + one halo frame per further window
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(got[k], one[k], equal_nan=True) and np.array_equal(got[k], ref[k], equal_nan=True), k
    if not ensemble:    # (the ensemble's one time stamp is that of the reference planner's LAST chunk -- quirk Q3 -- and follows the memory figure)
        assert np.array_equal(got.coords["time"], ref.coords["time"])
