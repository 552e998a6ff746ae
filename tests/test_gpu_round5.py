"""GPU tests of round 5: the chunk executor through the real engine, the bounded retention of borrowed ensemble chunks, ensemble
accumulation on a caller stream followed by finish without a synchronisation in between, the digest of the flagged records."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from pyorc_amd.synth import particle_stack

pytestmark = pytest.mark.gpu


class LazyStack:
    """What a dask-backed DataArray is to get_ffpiv: slicing along time is free, ``load()`` materialises (and takes a while)."""

    def __init__(self, data, seconds=0.0, log=None):
        self._data, self.seconds, self.log = data, seconds, log if log is not None else []
        self.dtype, self.shape = data.dtype, data.shape

    def __len__(self):
        return len(self._data)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return LazyStack(self._data[key], self.seconds, self.log)
        return self._data[key]

    def load(self):
        self.log.append(len(self._data))
        time.sleep(self.seconds)
        return np.array(self._data)      # a fresh array, like a materialised dask chunk


@pytest.mark.parametrize("ensemble", [False, True])
def test_lazy_chunks_prefetched_give_the_bits_of_the_serial_loop(gpu, ensemble):
    """get_piv over lazy chunks with the loads running ahead (depth 1 and 2) == the reference's serial order (depth 0) == the
    materialised stack, bit for bit; every chunk is loaded once; with loads as slow as the launches the run takes about the
    loads alone."""
    from pyorc_amd import executor, frames as F

    fr = particle_stack(101, 256, 320, seed=77)
    t = np.arange(101) / 30.0
    kw = dict(time=t, resolution=0.01, chunksize=26, ensemble_corr=ensemble)
    ref = F.get_piv(fr, 32, **kw)
    assert executor.LAST_STATS["depth"] == 0                       # a numpy stack: nothing to prefetch
    walls = {}
    for depth in (0, 1, 2):
        lazy = LazyStack(fr, seconds=0.05)
        t0 = time.perf_counter()
        got = F.get_piv(lazy, 32, prefetch=depth, **kw)
        walls[depth] = time.perf_counter() - t0
        assert lazy.log == [25, 25, 25, 26], lazy.log              # 101 frames in four pieces, no halo frame (round 6: resident stack)
        assert executor.LAST_STATS["depth"] == depth and executor.LAST_STATS["chunks"] == 4
        for k in ("v_x", "v_y", "corr", "s2n"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), (depth, k)
        assert np.array_equal(got.coords["time"], ref.coords["time"])
    # (what prefetching buys in wall time is measured where the clock is not shared with a GPU suite: tests/test_executor.py on
    # CPU, bench.py's lazy_host_chunks on the box; here only that the statistics are there)
    assert executor.LAST_STATS["load_s"] >= 4 * 0.05 * 0.9 and len(executor.LAST_STATS["waited_s_per_chunk"]) == 4 and walls[2] > 0


def test_borrowed_ensemble_chunks_respect_the_budget_and_a_chosen_mode(gpu, monkeypatch):
    """Ensemble.accumulate of DeviceFrames chunks borrows only when the caller has not chosen a mode, counts the
    borrowed bytes against LSPIV_ENSEMBLE_RETAIN_BYTES, and lets go of everything once the budget is exceeded (float32 fits,
    retain_complete False) instead of pinning every chunk until close()."""
    from pyorc_amd import DeviceFrames, piv

    fr = particle_stack(26, 128, 160, seed=5)
    chunk_bytes = fr.nbytes
    dev = DeviceFrames.from_host(fr)
    # (a) default: borrow, complete
    e = piv.Ensemble((128, 160), (32, 32), (16, 16))
    e.accumulate(dev, 0.2, 3.0)
    st = e.stats()
    assert len(e._held) == 1 and st["retain_complete"] and st["chunks_kept"] == 1 and st["bytes_kept"] >= chunk_bytes
    u0, v0, c0 = e.finish(0.2, 1)
    e.close()
    # (b) an explicit choice is respected: NONE keeps nothing and holds nothing
    e = piv.Ensemble((128, 160), (32, 32), (16, 16))
    e.set_retain(e.RETAIN_NONE)
    e.accumulate(dev, 0.2, 3.0)
    st = e.stats()
    assert e._held == [] and not st["retain_complete"] and st["chunks_kept"] == 0 and st["bytes_kept"] == 0
    u1, v1, c1 = e.finish(0.2, 1)
    assert np.array_equal(c0, c1) and np.allclose(u0, u1, atol=1e-3, equal_nan=True)
    e.close()
    # (c) COPY chosen: the handle owns copies, this object holds nothing
    e = piv.Ensemble((128, 160), (32, 32), (16, 16))
    e.set_retain(e.RETAIN_COPY)
    e.accumulate(dev, 0.2, 3.0)
    assert e._held == [] and e.stats()["retain_complete"] and e.stats()["chunks_kept"] == 1
    e.close()
    # (d) a budget of two and a half chunks: the third chunk exceeds it -> everything is let go, accumulation goes on
    # (the handle's first block of corr_max records is 1 MiB and counts as well)
    monkeypatch.setenv("LSPIV_ENSEMBLE_RETAIN_BYTES", str(int(2.5 * chunk_bytes) + (1 << 20) + 65536))
    e = piv.Ensemble((128, 160), (32, 32), (16, 16))
    for k in range(5):
        e.accumulate(dev, 0.2, 3.0)
        st = e.stats()
        if k < 2:
            assert st["retain_complete"] and len(e._held) == k + 1 and st["chunks_kept"] == k + 1
        else:
            assert not st["retain_complete"] and e._held == [] and st["chunks_kept"] == 0 and st["bytes_kept"] == 0
    u5, v5, c5 = e.finish(0.2, 1)
    assert np.array_equal(c5, 5 * c0) and e.stats()["float32_kept"] == e.stats()["flagged"]
    e.close()


def test_finish_waits_for_an_accumulate_on_a_caller_stream(gpu):
    """accumulate_dev on a caller stream, then finish (the library's stream) with NO synchronisation in between --
    the handle records an event per accumulating stream and its readers wait for it.  Large enough that the kernels are still
    running when finish is called; the result must equal the synchronous run's, bit for bit, every time."""
    from pyorc_amd import _lib, piv

    lib = _lib.load()
    H, W, T = 540, 960, 201
    n_win = 32 * 59
    d_f, d_o = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 8 * (T - 1) * n_win))
    _lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 99, 0.02))
    ref = None
    for trial in range(4):
        s = C.c_void_p()
        _lib.check(lib.lspiv_stream_create(C.byref(s)))
        e = piv.Ensemble((H, W), (32, 32), (16, 16))
        assert e.n_rows * e.n_cols == n_win
        e.set_retain(e.RETAIN_COPY if trial % 2 else e.RETAIN_BORROW)
        e.accumulate_dev(d_f.value, np.uint8, T, 0.2, 3.0, d_o.value, None, s.value if trial else None)
        if trial == 0:
            _lib.check(lib.lspiv_synchronize())
        out = e.finish(0.2, 1, return_mean=True)      # trial >= 1: straight after the asynchronous launch on stream s
        st = e.stats()
        assert st["retain_complete"] and st["rescued"] == st["flagged"]
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert np.array_equal(a, b, equal_nan=True), trial
        e.close()
        _lib.check(lib.lspiv_stream_destroy(s))
    lib.lspiv_dev_free(d_f); lib.lspiv_dev_free(d_o)


def test_flag_digest_names_the_list_not_its_length(gpu):
    """lspiv_ensemble_flag_digest: equal for handles holding the same sums (whatever order the kernel appended its records in), 0
    without records, and different when other windows are flagged."""
    from pyorc_amd import piv

    import pyorc_amd._lib as L

    L.set_option("rescue_kappa", 200000)     # an allowance large enough that some windows are flagged
    try:
        fr = particle_stack(8, 160, 224, seed=21, density=0.01)
        es = []
        for _ in range(2):
            e = piv.Ensemble((160, 224), (32, 32), (16, 16))
            e.accumulate(fr, 0.1, 1.0)
            es.append(e)
        n0, n1 = es[0].flag(0.1, 1), es[1].flag(0.1, 1)
        assert n0 == n1 and n0 > 1
        d0, d1 = es[0].flag_digest(), es[1].flag_digest()
        assert d0 == d1 and d0 != 0
        other = piv.Ensemble((160, 224), (32, 32), (16, 16))
        other.accumulate(particle_stack(8, 160, 224, seed=22, density=0.01), 0.1, 1.0)
        if other.flag(0.1, 1):
            assert other.flag_digest() != d0
        L.set_option("rescue", 0)                # the pass switched off: no records, digest 0
        assert es[0].flag(0.1, 1) == 0 and es[0].flag_digest() == 0
        for e in es + [other]:
            e.close()
    finally:
        L.set_option("rescue_kappa", 500)
        L.set_option("rescue", 1)


def test_project_hip_blocks_on_the_gpu(gpu, monkeypatch):
    """pyorc.project.project_hip (pyorc_amd.plugin) with the real projection kernel behind the apply_ufunc double: blocks of a graph
    share one device-resident plan, the result equals the numpy oracle's img_to_ortho bit for bit, also when the graph's blocks are
    run by several threads at once (dask's threaded scheduler)."""
    import sys
    import threading

    from oracle import project_oracle as pro
    from pyorc_amd import plugin
    from pyorc_amd.synth import projection_maps
    from tests import fake_xarray

    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    plugin.uninstall()       # no plans left over from other tests
    src, dst = (270, 480), (200, 360)
    maps = projection_maps(src, dst, tilt=0.15, seed=6)

    class CC:
        def map_idx_img_ortho(self, x, y, z):
            return maps[0], maps[1]

        def map_mean_idx_img_ortho(self, x, y, z):
            return maps[2], maps[3], maps[4]

    cam = particle_stack(24, src[0], src[1], seed=31)
    da = fake_xarray.DataArray(cam, ("time", "y", "x"), {"time": np.arange(24) / 25.0}, attrs={"_blocks": 4})
    y, x = np.arange(dst[0])[::-1] * 0.01, np.arange(dst[1]) * 0.01
    out = plugin.project_hip(da, CC(), x, y, 0.0, "mean")
    ref = pro.project_frames(cam, dst, *maps)
    assert out.values.dtype == np.float32 and np.array_equal(out.values.astype(np.float64), ref)
    assert len(plugin._PLANS) == 1
    # the blocks of ONE graph from four threads, each handed ANOTHER tuple around the same index maps (a real dask re-creates the
    # kwargs tuple per task): still the one plan of the graph (round 6: keyed by the arrays, not by the tuple)
    from pyorc_amd import executor

    dev = executor.current_device()
    res = [None] * 4
    def work(k):
        res[k] = plugin._project_block(cam[6 * k:6 * k + 6], plan_args=(maps[0], maps[1], maps[2], maps[3], maps[4]), dst_shape=dst, device=dev)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.array_equal(np.concatenate(res).astype(np.float64), ref) and len(plugin._PLANS) == 1
    plugin.uninstall()


@pytest.mark.parametrize("dtype", [np.uint8, np.float64])
def test_velocity_scaling_on_the_device_is_numpys(gpu, dtype):
    """get_piv scales u, v to m/s on the device before they cross PCIe when the resolution is a Python float (round 5:
    lspiv_piv_velocity_at / lspiv_scale_velocity_dev): bit for bit ``(u * res / dt[:, None, None]).astype(float32)`` as the reference
    writes it (ffpiv.py:418-419) -- uneven time steps, chunked, host frames and an HBM-resident stack; a numpy float64 resolution keeps
    numpy's own (float64-product) arithmetic on the host."""
    import pyorc_amd
    from pyorc_amd import DeviceFrames, frames as F

    fr = particle_stack(61, 160, 224, seed=14).astype(dtype)
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(0.02, 0.05, 61))
    dt = np.diff(t)
    res = 0.0123
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16))
    want_x = (u * res / np.expand_dims(dt, (1, 2))).astype(np.float32)
    want_y = (v * res / np.expand_dims(dt, (1, 2))).astype(np.float32)
    for stack in (fr, DeviceFrames.from_host(fr)):
        for cs in (None, 26):
            ds = F.get_piv(stack, 32, time=t, resolution=res, **({} if cs is None else {"chunksize": cs}))
            assert np.array_equal(ds["v_x"].view(np.uint32), want_x.view(np.uint32)) and np.array_equal(ds["v_y"].view(np.uint32), want_y.view(np.uint32))
            assert np.array_equal(ds["corr"], cm, equal_nan=True) and np.array_equal(ds["s2n"], sn, equal_nan=True)
    # the entry point itself, and the guard on the kind of resolution
    vx, vy, cm2, sn2 = pyorc_amd.piv_pairs(fr, (32, 32), (16, 16), scale=(res, res, dt))
    assert np.array_equal(vx.view(np.uint32), want_x.view(np.uint32)) and np.array_equal(vy.view(np.uint32), want_y.view(np.uint32))
    from pyorc_amd import piv, velocimetry as V

    assert piv.device_scaling_is_numpys(0.01, 1) and piv.device_scaling_is_numpys(np.float32(0.01), 0.5) and not piv.device_scaling_is_numpys(np.float64(0.01), 0.01)
    coords, _ = F.get_piv_coords((160, 224), (32, 32), (32, 32), (16, 16))
    ds64 = V.get_ffpiv(fr, coords["y"], coords["x"], dt, (32, 32), (16, 16), (32, 32), np.float64(res), np.float64(res), time=t)
    host_x = (u * np.float64(res) / np.expand_dims(dt, (1, 2))).astype(np.float32)       # numpy's arithmetic for that scalar type, whatever it is
    assert np.array_equal(ds64["v_x"].view(np.uint32), host_x.view(np.uint32))
    with pytest.raises(ValueError):
        pyorc_amd.piv_pairs(fr, (32, 32), (16, 16), scale=(res, res, dt[:-1]))
