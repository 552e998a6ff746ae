"""The chunk executor of get_ffpiv (pyorc_amd/executor.py): lazy chunks are materialised ahead of the launches that consume them.

Reference loop: pyorc/velocimetry/ffpiv.py:13-21 (load_frame_chunk with its TypeError retry), :348-370 and :399-440 (load, then compute,
chunk after chunk).  CPU tests: the GPU call is replaced by the oracle (plus a sleep where a test measures the overlap)."""
import importlib
import os
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyorc_amd import executor  # noqa: E402
from pyorc_amd.velocimetry import load_frame_chunk  # noqa: E402


class LazyChunk:
    """What a dask-backed ``frames[a:b]`` is to the executor: ``len``, slicing, and a ``load()`` that does the work (here: sleeps --
    like dask's schedulers and numpy it releases the GIL while it does)."""

    log = []

    def __init__(self, data, seconds=0.0, fail=None, name=None):
        self.data, self.seconds, self.fail, self.name = data, seconds, fail, name
        self.loads = 0

    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        return LazyChunk(self.data[key], self.seconds, None if self.fail is TypeError else self.fail, self.name)

    def load(self):
        self.loads += 1
        LazyChunk.log.append(("load-start", self.name, threading.current_thread().name))
        time.sleep(self.seconds)
        if self.fail is not None:
            raise self.fail("synthetic failure of " + str(self.name))
        LazyChunk.log.append(("load-end", self.name, threading.current_thread().name))
        return self.data


def test_chunks_come_in_order_and_are_loaded_once():
    chunks = [LazyChunk(np.full(3, n), name=n) for n in range(7)]
    for depth in (0, 1, 3, 10):
        for c in chunks:
            c.loads = 0
        got = [(n, int(a[0])) for n, a in executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=depth)]
        assert got == [(n, n) for n in range(7)]
        assert all(c.loads == 1 for c in chunks)


def test_depth_zero_is_the_serial_loop_on_the_callers_thread():
    LazyChunk.log = []
    chunks = [LazyChunk(np.zeros(2), name=n) for n in range(3)]
    me = threading.current_thread().name
    for n, _ in executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=0):
        LazyChunk.log.append(("compute", n, me))
    assert [e[:2] for e in LazyChunk.log] == [("load-start", 0), ("load-end", 0), ("compute", 0), ("load-start", 1), ("load-end", 1),
                                             ("compute", 1), ("load-start", 2), ("load-end", 2), ("compute", 2)]
    assert all(e[2] == me for e in LazyChunk.log)


def test_load_overlaps_compute_wall_is_max_not_sum():
    """Six chunks, load 0.12 s, compute 0.12 s: the serial loop takes 12 x 0.12 s, one-deep prefetch 7 x 0.12 s."""
    n_chunks, load_s, compute_s = 6, 0.12, 0.12

    def run(depth):
        chunks = [LazyChunk(np.zeros(2), load_s, name=n) for n in range(n_chunks)]
        ex = executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=depth)
        t0 = time.perf_counter()
        for _n, _a in ex:
            time.sleep(compute_s)     # the launch: ctypes releases the GIL, like sleep
        return time.perf_counter() - t0, ex.stats

    serial, st0 = run(0)
    overlapped, st1 = run(1)
    assert serial >= n_chunks * (load_s + compute_s) * 0.98
    ideal = load_s + n_chunks * max(load_s, compute_s)
    assert overlapped <= ideal * 1.25, (overlapped, ideal)
    assert overlapped <= 0.72 * serial, (overlapped, serial)
    # the statistics say the same: every load took its time, but the consumer only waited for the first
    assert st1["load_s"] >= n_chunks * load_s * 0.98 and st1["waited_s"] <= 2.5 * load_s
    assert st0["waited_s"] >= n_chunks * load_s * 0.98


def test_slow_loader_bounds_the_run_and_depth_bounds_the_memory():
    """Loads slower than the launches: wall ~ the loads alone; never more than depth + 1 loaded chunks alive."""
    alive, peak = [0], [0]
    lock = threading.Lock()

    class Counted(LazyChunk):
        def load(self):
            out = super().load()
            with lock:
                alive[0] += 1
                peak[0] = max(peak[0], alive[0])
            return out

    chunks = [Counted(np.zeros(2), 0.05, name=n) for n in range(8)]
    t0 = time.perf_counter()
    for _n, _a in executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=2):
        time.sleep(0.01)
        with lock:
            alive[0] -= 1
    wall = time.perf_counter() - t0
    assert wall <= 8 * 0.05 * 1.3 + 0.05
    assert peak[0] <= 3


def test_type_error_retry_of_the_reference_survives_the_thread():
    """ffpiv.py:17-21: a chunk whose load raises TypeError is retried without its last frame -- on the worker thread too."""
    data = np.arange(5)
    chunks = [LazyChunk(np.arange(4), name=0), LazyChunk(data, fail=TypeError, name=1), LazyChunk(np.arange(3), name=2)]
    got = [a for _, a in executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=1)]
    assert [len(a) for a in got] == [4, 4, 3] and np.array_equal(got[1], data[:-1])


def test_loader_exception_surfaces_at_its_chunk_and_cancels_the_rest():
    chunks = [LazyChunk(np.zeros(2), 0.01, name=n) for n in range(6)]
    chunks[2] = LazyChunk(np.zeros(2), 0.01, fail=RuntimeError, name=2)
    seen = []
    ex = executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=1)
    with pytest.raises(RuntimeError, match="synthetic failure of 2"):
        for n, _ in ex:
            seen.append(n)
    assert seen == [0, 1]
    assert chunks[4].loads == 0 and chunks[5].loads == 0       # never started
    assert not any(t.name.startswith("lspiv-load") for t in threading.enumerate())


def test_leaving_the_loop_early_releases_the_threads():
    chunks = [LazyChunk(np.zeros(2), 0.01, name=n) for n in range(6)]
    for n, _ in executor.ChunkPrefetcher(chunks, load_frame_chunk, depth=2):
        if n == 1:
            break
    time.sleep(0.05)
    assert not any(t.name.startswith("lspiv-load") for t in threading.enumerate())
    assert chunks[5].loads == 0


def test_environment_default(monkeypatch):
    monkeypatch.delenv("LSPIV_PREFETCH_DEPTH", raising=False)
    assert executor.default_depth() == 1
    monkeypatch.setenv("LSPIV_PREFETCH_DEPTH", "0")
    assert executor.default_depth() == 0 and executor.ChunkPrefetcher([], load_frame_chunk).depth == 0
    monkeypatch.setenv("LSPIV_PREFETCH_DEPTH", "-1")
    with pytest.raises(ValueError):
        executor.default_depth()



def _mirrors_with_fake_xarray(monkeypatch):
    from oracle import c_oracle
    from tests import fake_xarray

    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    import pyorc_amd.frames as F
    import pyorc_amd.velocimetry as V

    for mod in (V, F):
        importlib.reload(mod)
    monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
    return F, V, fake_xarray, c_oracle


def test_get_ffpiv_gives_the_same_dataset_with_and_without_prefetch(monkeypatch):
    """Per time step and ensemble, through get_piv on a (double of a) lazy DataArray: depth 0, 1 and 3 give identical results,
    every chunk is loaded exactly once, the launches see the chunks in order, and with a slow load + slow launch the prefetched
    run takes about half the serial one."""
    from pyorc_amd.synth import particle_stack
    from tests.test_shard_gloo import OracleEnsemble

    F, V, fake_xarray, c_oracle = _mirrors_with_fake_xarray(monkeypatch)
    try:
        load_s = compute_s = 0.06
        loads, order = [], []

        class Lazy(fake_xarray.DataArray):
            data = property(lambda self: None)      # not in memory

            def __getitem__(self, key):
                sub = super().__getitem__(key)
                if isinstance(key, slice):
                    sub.__class__ = Lazy
                return sub

            def load(self):
                loads.append(float(self.coords["time"].values[0]))
                time.sleep(load_s)
                return super().load()

        def fake_pairs(fr, ws, ov, thr=None, pair_offset=0, out=None, scale=None):
            from tests.doubles import oracle_piv_pairs

            order.append(pair_offset)
            time.sleep(compute_s)
            return oracle_piv_pairs(fr, ws, ov, thr, pair_offset, out, scale)

        class SlowEnsemble(OracleEnsemble):
            def accumulate(self, frames, corr_min, s2n_min, thr=None, out=None):
                time.sleep(compute_s)
                return super().accumulate(np.asarray(frames), corr_min, s2n_min, thr, out)

            def finish(self, count_min, n_frames):
                mean = self.s / np.maximum(self.k, 1)[:, None, None]
                u, v = self.po.u_v_displacement(mean[None], self.n_rows, self.n_cols)
                return u.astype(np.float32), v.astype(np.float32), self.k.astype(np.float32)

            def close(self):
                pass

        monkeypatch.setattr(V.piv, "piv_pairs", fake_pairs)
        monkeypatch.setattr(V.piv, "Ensemble", SlowEnsemble)
        monkeypatch.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 5)
        __import__("tests.doubles", fromlist=["x"]).use_host_stacks(monkeypatch)   # the lazy path keeps the stack "resident": numpy here
        fr = particle_stack(31, 64, 96, seed=11)
        t = np.arange(31) / 25.0
        da = Lazy(fr, ("time", "y", "x"), {"time": t, "y": np.arange(64)[::-1] * 0.02, "x": np.arange(96) * 0.02})
        for ens in (False, True):
            results, walls = {}, {}
            for depth in (0, 1, 3):
                loads.clear(); order.clear()
                t0 = time.perf_counter()
                ds = F.get_piv(da, 32, resolution=0.02, chunksize=6, ensemble_corr=ens, prefetch=depth)
                walls[depth] = time.perf_counter() - t0
                results[depth] = {k: np.array(ds[k].values) for k in ("v_x", "v_y", "corr", "s2n")}
                assert len(loads) == 6 and loads == sorted(loads), loads          # 31 frames in six pieces of 5 (the last: 6), each loaded once
                if not ens:
                    assert order == [0, 5, 10, 15, 20]      # launches start on the anchors (the last one spans two: frames 25 .. 30 arrive together)
                st = executor.LAST_STATS
                assert st["depth"] == depth and st["chunks"] == 6
            for depth in (1, 3):
                for k, ref in results[0].items():
                    assert np.array_equal(results[depth][k], ref, equal_nan=True), (ens, depth, k)
            # serial: 6 x (load + compute); prefetched: load + 6 x compute (+ the oracle's own time in both)
            assert walls[1] <= walls[0] - 3 * load_s, (ens, walls)       # five of the six loads hide behind the launches (give one for scheduling noise)
        # an already materialised stack has nothing to prefetch: no thread is started
        loads.clear()
        F.get_piv(fr, 32, time=t, resolution=0.02, chunksize=6)
        assert executor.LAST_STATS["depth"] == 0
    finally:
        monkeypatch.undo()
        import pyorc_amd.frames as F2
        import pyorc_amd.velocimetry as V2

        for mod in (V2, F2):
            importlib.reload(mod)


def test_a_lazy_stack_without_xarray_stays_lazy_in_get_piv(monkeypatch):
    """get_piv on a plain (non-xarray) stack whose time slices materialise on ``load()``: it must NOT be converted to one numpy array
    up front (that would load every frame before the first launch) but reach get_ffpiv lazy, chunk by chunk, ahead of the launches."""
    from oracle import c_oracle
    from pyorc_amd import frames as F, velocimetry as V
    from pyorc_amd.synth import particle_stack

    loads = []

    class Lazy:
        def __init__(self, data):
            self._d, self.dtype, self.shape = data, data.dtype, data.shape

        def __len__(self):
            return len(self._d)

        def __getitem__(self, key):
            return Lazy(self._d[key]) if isinstance(key, slice) else self._d[key]

        def load(self):
            loads.append(len(self._d))
            return np.array(self._d)

    monkeypatch.setattr(V.piv, "piv_pairs", __import__("tests.doubles", fromlist=["x"]).oracle_piv_pairs)
    monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
    monkeypatch.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 5)
    stacks = __import__("tests.doubles", fromlist=["x"]).use_host_stacks(monkeypatch)
    fr = particle_stack(16, 64, 96, seed=2)
    t = np.arange(16) / 25.0
    ref = F.get_piv(fr, 32, time=t, resolution=0.02, chunksize=6)
    got = F.get_piv(Lazy(fr), 32, time=t, resolution=0.02, chunksize=6)
    # three pieces, NO halo frame between them (every frame is loaded exactly once), uploaded to their places in the resident stack
    assert loads == [5, 5, 6] and stacks.uploads == [(0, 5), (5, 5), (10, 6)]
    assert executor.LAST_STATS["adaptive"] and executor.LAST_STATS["chunks"] == 3
    for k in ref:
        assert np.array_equal(got[k], ref[k], equal_nan=True)


def test_a_chunk_that_lost_its_last_frame_leaves_no_gap_in_the_result(monkeypatch):
    """load_frame_chunk's TypeError retry (ffpiv.py:17-21) shortens a piece by a frame.  The lazy path loads every frame ONCE (no halo),
    so a frame a loader drops is absent, and so are the two pairs it belongs to; the pieces before and after run as they are.  The run's
    result arrays are allocated for every pair and written slice by slice; the missing pairs must not show up as holes of
    uninitialised memory -- the result holds exactly the delivered pairs, in order, with their time stamps."""
    from pyorc_amd import frames as F, velocimetry as V
    from pyorc_amd.synth import particle_stack
    from tests.doubles import oracle_piv_pairs, use_host_stacks

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
            if self.lo == 5 and len(self._d) == 5:      # the second piece (frames 5 .. 9) fails at full length: frame 9 cannot be decoded
                raise TypeError("cannot decode the last frame of this block")
            return np.array(self._d)

    monkeypatch.setattr(V.piv, "piv_pairs", oracle_piv_pairs)
    monkeypatch.setattr(V.window, "available_memory", lambda: 1e12)
    monkeypatch.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 5)
    use_host_stacks(monkeypatch)
    fr = particle_stack(16, 64, 96, seed=2)
    t = np.arange(16) / 25.0
    ref = F.get_piv(fr, 32, time=t, resolution=0.02, chunksize=6)             # pairs 0 .. 14
    got = F.get_piv(Lazy(fr), 32, time=t, resolution=0.02, chunksize=6)       # pairs 8 (frames 8, 9) and 9 (frames 9, 10) are lost with frame 9
    keep = [p for p in range(15) if p not in (8, 9)]
    assert got["v_x"].shape[0] == 13 and np.array_equal(got.coords["time"], t[1:][keep])
    for k in ref:
        assert np.array_equal(got[k], ref[k][keep], equal_nan=True), k
