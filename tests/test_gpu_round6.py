"""GPU tests of round 6: lazy stacks resident in HBM (lspiv_upload_frames + launches on the anchors) against the materialised stack,
the hand-off from project_hip to get_ffpiv on the device, projection blocks overlapping a PIV host call (library-recorded HIP events),
and the Ensemble's borrowed chunks across a change of the retention mode."""
import ctypes as C
import sys
import threading
import time

import numpy as np
import pytest

from pyorc_amd.synth import particle_stack, projection_maps

pytestmark = pytest.mark.gpu


class LazyStack:
    """What a dask-backed DataArray is to get_ffpiv: slicing along time is free, ``load()`` materialises."""

    def __init__(self, data, log=None, blocks=None):
        self._data, self.log = data, log if log is not None else []
        self.dtype, self.shape = data.dtype, data.shape
        if blocks:
            self.chunks = (tuple(blocks),) + tuple((n,) for n in data.shape[1:])

    def __len__(self):
        return len(self._data)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return LazyStack(self._data[key], self.log)
        return self._data[key]

    def load(self):
        self.log.append(len(self._data))
        return np.array(self._data)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
@pytest.mark.parametrize("ensemble", [False, True])
def test_lazy_stack_runs_resident_with_default_arguments(gpu, dtype, ensemble):
    """get_piv over a lazy stack with NO chunksize / prefetch argument: several loads (host budget and overlap granule; cut on the
    stack's own blocks), every frame loaded once, launches on the anchors -- and the bits of the materialised stack, for the three
    frame dtypes pyorc hands over.  float64 rides on a DC offset large enough for the narrowing guard to act."""
    from pyorc_amd import executor, frames as F

    fr = particle_stack(131, 256, 320, seed=31).astype(dtype)
    if dtype == np.float64:
        fr = fr + 4096.0 * (1 + np.arange(131) % 3)[:, None, None]
    t = np.arange(131) / 30.0
    kw = dict(time=t, resolution=0.01, ensemble_corr=ensemble)
    ref = F.get_piv(fr, 32, **kw)
    lazy = LazyStack(fr, blocks=[20] * 6 + [11])
    got = F.get_piv(lazy, 32, **kw)
    st = executor.LAST_STATS
    assert st["plan"]["source"] == "frames" and st["adaptive"] and st["chunks"] == len(lazy.log) >= 4
    assert sum(lazy.log) == 131 and all(n % 20 == 0 for n in lazy.log[:-1])          # every frame once, whole blocks
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(got[k], ref[k], equal_nan=True), (dtype, ensemble, k)
    assert np.array_equal(got.coords["time"], ref.coords["time"])


def test_upload_frames_is_the_staging_of_the_piv_host_entry_point(gpu):
    """DeviceFrames.upload (lspiv_upload_frames) in pieces + one launch on the resident stack == lspiv_piv_pairs on the host stack,
    bit for bit: uint8, float32 and float64 (narrowed, with and without frames that trigger the DC-offset guard)."""
    from pyorc_amd import DeviceFrames, piv

    base = particle_stack(41, 192, 256, seed=3)
    for dtype, offset, thr in ((np.uint8, 0, None), (np.float32, 0, None), (np.float64, 0, None), (np.float64, 1e5, None), (np.float64, 1e5, 0.1)):
        fr = base.astype(dtype)
        if offset:
            fr = fr + offset * (np.arange(41) % 2)[:, None, None]
        ref = piv.piv_pairs(fr, (32, 32), (16, 16), thr)
        d = DeviceFrames.empty(fr.shape, DeviceFrames.device_dtype(dtype))
        for f0, f1 in ((0, 7), (7, 8), (8, 30), (30, 41)):
            assert d.upload(f0, fr[f0:f1], thr) == f1 - f0
        got = piv.piv_pairs(d, (32, 32), (16, 16), thr)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True), (dtype, offset, thr)
    with pytest.raises(TypeError):
        DeviceFrames.empty((4, 192, 256), np.uint8).upload(0, base[:2].astype(np.float64))
    with pytest.raises(IndexError):
        DeviceFrames.empty((4, 192, 256), np.uint8).upload(3, base[:2])


def _ortho_lazy(cam, maps, dst, block=20, extra_layer=False, xarray_mod=None):
    from pyorc_amd import plugin
    from tests import lazy_doubles

    video = lazy_doubles.from_frames(cam, block=block)
    ortho = lazy_doubles.frames_project(video, maps, dst, plugin.project_hip)
    return (ortho.map_time(lambda blk: blk, "astype") if extra_layer else ortho), video


@pytest.mark.parametrize("reducer_mean,recipe_frames", [(True, False), (False, False), (True, True)])
def test_project_hip_product_stays_in_hbm_for_get_piv(gpu, monkeypatch, reducer_mean, recipe_frames):
    """frames.project(method="hip") -> [fillna] -> get_piv(engine="hip"): the camera blocks are uploaded and projected into the
    resident stack on the device; one more layer in between and the projected blocks come back to the host and go up again.  Same bits
    both ways, and those of Projection.project_frames + get_piv on a numpy stack."""
    from pyorc_amd import executor, frames as F, plugin
    from pyorc_amd.project import Projection
    from tests import lazy_doubles

    monkeypatch.setitem(sys.modules, "xarray", lazy_doubles)
    plugin.uninstall()
    src, dst = (270, 480), (200, 360)
    maps = projection_maps(src, dst, tilt=0.1, seed=2)
    if not reducer_mean:
        maps = (maps[0], maps[1], None, None, None)
    cam = particle_stack(91, src[0], src[1], seed=8)
    if recipe_frames:
        # what the reference's recipe hands to project: normalize -> edge_detect -> minmax come first (examples/ngwerere/ngwerere.yml:5-11),
        # signed float32 camera frames -- uploaded as they are and projected through the float32 tiles
        from pyorc_amd import filters
        cam = filters.minmax(filters.edge_detect(filters.normalize(cam, 15), 1, 2), -5, 5)
        assert cam.dtype == np.float32 and cam.min() < 0
    t = np.arange(91) / 30.0
    kw = dict(time=t, resolution=0.01)
    try:
        ortho, video = _ortho_lazy(cam, maps, dst)
        got = F.get_piv(ortho, 32, **kw)
        st = dict(executor.LAST_STATS)
        assert st["plan"]["source"] == "camera" and st["chunks"] >= 3
        calls = {k: v for k, v in video.calls.items() if k[0].startswith("project_block")}
        assert calls == {}                                                  # no projected block was ever computed for the host
        other, video2 = _ortho_lazy(cam, maps, dst, extra_layer=True)
        ref = F.get_piv(other, 32, **kw)
        assert executor.LAST_STATS["plan"]["source"] == "frames"
        calls = {k[1]: v for k, v in video2.calls.items() if k[0].startswith("project_block")}
        assert calls == {i: 1 for i in range(5)}                            # 91 frames in blocks of 20: each projected once
        p = Projection(src, dst, *[m for m in maps])
        whole = F.get_piv(p.project_frames(cam), 32, **kw)
        p.close()
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(got[k], ref[k], equal_nan=True) and np.array_equal(got[k], whole[k], equal_nan=True), k
        e1 = F.get_piv(ortho, 32, ensemble_corr=True, **kw)
        e2 = F.get_piv(other, 32, ensemble_corr=True, **kw)
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(e1[k], e2[k], equal_nan=True), k
    finally:
        plugin.uninstall()


def _trace_read(lib):
    n = C.c_int64(0)
    cap = 256
    kind = (C.c_int32 * cap)()
    t0 = (C.c_double * cap)()
    t1 = (C.c_double * cap)()
    assert lib.lspiv_trace_read(cap, kind, t0, t1, C.byref(n)) == 0
    return [(int(kind[i]), float(t0[i]), float(t1[i])) for i in range(min(cap, n.value))]


def test_a_projection_block_runs_inside_a_concurrent_piv_host_call(gpu):
    """lspiv_project_frames no longer takes the `host` lock that lspiv_piv_pairs holds from its first upload to its
    last download, nor its stream or workspaces.  Evidence is the library's own HIP events, not a wall clock: a projection kernel's
    span lies INSIDE the span of a PIV host call issued by another thread on the same device; and the results of both are those of
    the calls made one after the other."""
    from pyorc_amd import _lib, piv
    from pyorc_amd.project import Projection

    lib = gpu
    fr = particle_stack(120, 1080, 1920, seed=1).astype(np.float64)      # 2 GB of float64: ~50 ms of staging + upload
    src, dst = (540, 960), (400, 720)
    plan = Projection(src, dst, *projection_maps(src, dst, tilt=0.1, seed=5))
    cam = particle_stack(20, src[0], src[1], seed=2)
    ref_piv = piv.piv_pairs(fr, (32, 32), (16, 16))
    ref_proj = plan.project_frames(cam)
    inside = 0
    for attempt in range(4):
        assert lib.lspiv_trace(1) == 0
        out = {}
        th = threading.Thread(target=lambda: out.setdefault("piv", piv.piv_pairs(fr, (32, 32), (16, 16))))
        th.start()
        time.sleep(0.01)
        for _ in range(6):
            out["proj"] = plan.project_frames(cam)
        th.join()
        spans = _trace_read(lib)
        assert lib.lspiv_trace(0) == 0
        pivs = [s for s in spans if s[0] == 0]
        projs = [s for s in spans if s[0] == 1]
        assert len(pivs) == 1 and len(projs) == 6 and all(b >= a for _, a, b in spans)
        inside = sum(1 for _, a, b in projs if pivs[0][1] < a and b < pivs[0][2])
        for a, b in zip(out["piv"], ref_piv):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(out["proj"], ref_proj)
        if inside:
            break
    assert inside >= 1, (pivs, projs)
    # two projection calls from two threads use the two slots: their kernel spans may overlap each other, results unchanged
    assert lib.lspiv_trace(1) == 0
    res = [None, None]
    ts = [threading.Thread(target=lambda i=i: res.__setitem__(i, plan.project_frames(cam))) for i in range(2)]
    [x.start() for x in ts]; [x.join() for x in ts]
    assert lib.lspiv_trace(0) == 0
    assert np.array_equal(res[0], ref_proj) and np.array_equal(res[1], ref_proj)
    plan.close()


def test_trace_is_off_by_default_and_empty_after_stop(gpu):
    from pyorc_amd import piv

    lib = gpu
    piv.piv_pairs(particle_stack(3, 128, 160, seed=1), (32, 32), (16, 16))
    assert _trace_read(lib) == []
    assert lib.lspiv_trace(1) == 0
    piv.piv_pairs(particle_stack(3, 128, 160, seed=1), (32, 32), (16, 16))
    sp = _trace_read(lib)
    assert len(sp) == 1 and sp[0][0] == 0 and 0 <= sp[0][1] <= sp[0][2]
    assert lib.lspiv_trace(0) == 0 and _trace_read(lib) == []


def test_borrowed_chunks_stay_pinned_across_a_change_of_the_retention_mode(gpu):
    """BORROW, then set_retain(COPY) or set_retain(NONE) + finish: the handle still holds the borrowed pointers
    (lspiv_ensemble_set_retain only stores the mode), so the Python object must keep the stacks alive until close() -- it used to drop
    them, and the final fit's float64 rescue read freed HBM."""
    from pyorc_amd import DeviceFrames, piv

    fr = particle_stack(26, 128, 160, seed=5)
    for later in (piv.Ensemble.RETAIN_COPY, piv.Ensemble.RETAIN_NONE):
        e = piv.Ensemble((128, 160), (32, 32), (16, 16))
        e.accumulate(DeviceFrames.from_host(fr), 0.2, 3.0)           # borrowed; the only other reference dies with this statement
        assert len(e._held) == 1
        e.set_retain(later)
        assert len(e._held) == 1, "the borrowed stack was let go while the handle still points at it"
        # churn the allocator: if the stack had been freed, these would land on its bytes
        junk = [DeviceFrames.from_host(np.full_like(fr, 255 - k)) for k in range(4)]
        u, v, c = e.finish(0.2, 1)
        ref = piv.Ensemble((128, 160), (32, 32), (16, 16))
        ref.accumulate(fr, 0.2, 3.0)
        ur, vr, cr = ref.finish(0.2, 1)
        assert np.array_equal(c, cr) and np.array_equal(u, ur, equal_nan=True) and np.array_equal(v, vr, equal_nan=True)
        del junk
        e.close(); ref.close()
        assert e._held == []
