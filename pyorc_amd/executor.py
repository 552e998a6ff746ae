"""Chunk executor of ``get_ffpiv``: the lazy chunks of the frame stack are materialised AHEAD of the launches that consume them.

The reference's loops (pyorc/velocimetry/ffpiv.py:348-370 ensemble, :399-440 per time step) are strictly serial::

    for n in range(len(frames_chunks)):
        da = load_frame_chunk(frames_chunks[n])     # dask executes video decode + orthoprojection + filters HERE
        ... ffpiv.cross_corr(da.values) ...         # then the arithmetic

With ``ffpiv`` on host cores both halves compete for the same cores and the order hardly matters.  With the arithmetic on the GPU
(6 ms per 1000 pairs at 1080p) the ``.load()`` of the next chunk IS the wall time of a run, and a serial loop leaves the GPU idle during
it and the host idle during the launch (upload over PCIe + kernel + download).  :class:`ChunkPrefetcher` runs
``load_frame_chunk(frames_chunks[n + 1 ... n + depth])`` on worker threads while the caller's thread launches chunk ``n`` -- dask's
schedulers and numpy release the GIL, and so does every ctypes call into ``liblspiv_hip.so`` --, hands the chunks over IN ORDER, and
re-raises a loader's exception at the position of its chunk, i.e. exactly where the serial loop would have raised it.  The results
are those of the serial loop, bit for bit: only WHEN a chunk is materialised changes, not what is computed from it.

How far ahead (round 6).  ``depth`` chunks may be loaded or loading beyond the one the consumer holds, so ``depth + 1`` chunks of frames
sit in host memory: the planner of ``get_ffpiv`` sizes a lazy chunk as (host budget) / (``max_depth()`` + 1) for exactly that reason.
Unless the caller fixes it (``prefetch=`` / ``LSPIV_PREFETCH_DEPTH``), the depth ADAPTS to what the run shows: after the first chunk,
``ceil(load time / launch time)`` loads run side by side; afterwards one more whenever the consumer still waited for its chunk, one
fewer (for good) when the extra loader did not make the chunks arrive faster -- a dask ``.load()`` that already uses every core gains
nothing from a second one next to it, a single-threaded decoder gains almost linearly.  Never more than ``max_depth()``.

Worker threads inherit the caller's HIP device (``hipSetDevice`` is per thread): a loader that itself calls the library -- a
``project_hip`` dask block, a host-pointer filter -- works on the GPU its caller selected, not on device 0.
"""

from __future__ import annotations

import math
import os
import threading
import time as _time
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

DEFAULT_DEPTH = 1          # the depth an adaptive run starts from
DEFAULT_WORKERS = 1
DEFAULT_MAX_DEPTH = 4


def _env_int(name: str, lo: int) -> Optional[int]:
    v = os.environ.get(name)
    if v is None or v == "":
        return None
    d = int(v)
    if d < lo:
        raise ValueError(f"{name} must be >= {lo}, got {d}")
    return d


def default_depth() -> int:
    """Prefetch depth when the caller does not say and adaptation is off: ``LSPIV_PREFETCH_DEPTH`` (0 = the reference's serial loop), else 1."""
    d = _env_int("LSPIV_PREFETCH_DEPTH", 0)
    return DEFAULT_DEPTH if d is None else d


def default_workers() -> int:
    """Loader threads of a FIXED-depth run when the caller does not say: ``LSPIV_PREFETCH_WORKERS``, else 1 (a dask ``.load()`` is parallel
    inside; more than one thread pays for loaders that run on one core each and needs ``depth >= workers``).  An adaptive run has
    ``max_depth()`` threads and as many loads in flight as its current depth."""
    w = _env_int("LSPIV_PREFETCH_WORKERS", 1)
    return DEFAULT_WORKERS if w is None else w


def max_depth() -> int:
    """The deepest an adaptive run goes (``LSPIV_PREFETCH_MAX_DEPTH``, default 4): ``max_depth() + 1`` chunks of frames is what the planner
    of ``get_ffpiv`` budgets host memory for."""
    d = _env_int("LSPIV_PREFETCH_MAX_DEPTH", 1)
    return DEFAULT_MAX_DEPTH if d is None else d


def adaptive_by_default() -> bool:
    """Adaptation is the default unless the environment fixes the depth (``LSPIV_PREFETCH_DEPTH``)."""
    return _env_int("LSPIV_PREFETCH_DEPTH", 0) is None


def current_device() -> Optional[int]:
    """The calling thread's HIP device, or None when the library is not built / there is no device (CPU tests, oracle doubles)."""
    try:
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        n = C.c_int(0)
        if lib.lspiv_device_count(C.byref(n)) != 0 or n.value < 1:
            return None
        d = C.c_int(0)
        return int(d.value) if lib.lspiv_get_device(C.byref(d)) == 0 else None
    except Exception:
        return None


def bind_device(device: Optional[int]) -> None:
    """Make ``device`` the calling thread's HIP device (no-op for None)."""
    if device is None:
        return
    from . import _lib

    _lib.check(_lib.load().lspiv_set_device(int(device)))


class ChunkPrefetcher:
    """Iterate ``(n, load(chunks[n]))`` in order, with up to ``depth`` loads running ahead on worker threads.

    * ``chunks``: the lazy chunks (``frames[a:b]`` slices of an ``xr.DataArray``); the list is NOT modified, but the prefetcher
      drops its own reference to a chunk as soon as its load has been handed over.
    * ``load``: ``load_frame_chunk`` of ``pyorc_amd.velocimetry`` (``da.load()`` with the reference's ``TypeError`` retry).
    * ``depth``: how many chunks may be loaded (or loading) beyond the one the consumer holds.  0 = load on the caller's
      thread when the chunk is asked for: the reference's order of events.  ``None``: ``LSPIV_PREFETCH_DEPTH`` when set, else 1.
    * ``adaptive``: let the depth follow the run (module docstring); ``depth`` is then where it starts.  Default: on when neither the
      caller (``depth=``) nor the environment fixed the depth.
    * ``workers``: loader threads of a fixed-depth run (default 1: loads run one after the other, in chunk order, like the serial
      loop's); an adaptive run owns ``max_depth`` threads and keeps ``depth`` loads in flight.
    * ``device``: HIP device the worker threads select first thing (default: the creating thread's).
    * Errors: an exception raised by ``load(chunks[k])`` is re-raised by the iterator when chunk ``k`` is due, after chunks
      ``0 .. k-1`` have been handed over; loads queued behind it are cancelled.  Leaving the loop early (``break``, an
      exception in the consumer, ``close()``, leaving a ``with`` block) cancels what has not started and waits for what has.

    ``stats`` after (or during) the run: per chunk the seconds its load took and the seconds the consumer waited for it; the
    wall time the consumer spent waiting is what prefetching could NOT hide.
    """

    def __init__(self, chunks: Sequence, load: Callable, depth: Optional[int] = None, workers: Optional[int] = None,
                 adaptive: Optional[bool] = None, max_depth_: Optional[int] = None, device: Optional[int] = -1):
        self._chunks: List = list(chunks)
        self._load = load
        if adaptive is None:
            adaptive = depth is None and adaptive_by_default()
        self.depth = default_depth() if depth is None else int(depth)
        if self.depth < 0:
            raise ValueError(f"prefetch depth must be >= 0, got {self.depth}")
        self.adaptive = bool(adaptive) and self.depth > 0
        self.max_depth = max(self.depth, max_depth() if max_depth_ is None else int(max_depth_)) if self.adaptive else self.depth
        if self.adaptive:
            self.workers = self.max_depth if workers is None else max(1, int(workers))
        else:
            self.workers = default_workers() if workers is None else max(1, int(workers))
        self.device = current_device() if device == -1 else device
        self._pool: Optional[ThreadPoolExecutor] = None
        self._futures: dict = {}
        self._next_submit = 0
        self._lock = threading.Lock()
        self.load_s: List[float] = [0.0] * len(self._chunks)
        self.wait_s: List[float] = [0.0] * len(self._chunks)
        self.consume_s: List[float] = [0.0] * len(self._chunks)
        self.depth_history: List[int] = []
        self._closed = False
        # adaptation state
        self._frozen = False
        self._win_start_n = 0                 # first chunk of the current evaluation window
        self._win_start_t = 0.0
        self._prev_rate: Optional[float] = None   # chunks / s at depth - 1, when the depth was just raised

    def __len__(self) -> int:
        return len(self._chunks)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- loader side ------------------------------------------------------------------------------------------------------------
    def _init_worker(self) -> None:
        try:
            bind_device(self.device)
        except Exception:       # a worker without a device still loads; whatever needs the GPU in it raises its own error
            pass

    def _timed_load(self, n: int):
        t0 = _time.perf_counter()
        try:
            return self._load(self._chunks[n])
        finally:
            self.load_s[n] = _time.perf_counter() - t0

    def _submit_up_to(self, last: int) -> None:
        """Queue the loads of chunks ``_next_submit .. last`` (inclusive)."""
        if self.depth == 0:
            return
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="lspiv-load", initializer=self._init_worker)
        last = min(last, len(self._chunks) - 1)
        while self._next_submit <= last:
            n = self._next_submit
            self._futures[n] = self._pool.submit(self._timed_load, n)
            self._next_submit += 1

    # -- adaptation -------------------------------------------------------------------------------------------------------------
    def _adapt(self, n: int, now: float) -> None:
        """Called when the consumer comes back for chunk ``n + 1``: chunk ``n`` has been loaded, waited for and processed."""
        if not self.adaptive or self._frozen:
            return
        if n == 0:
            # first estimate: as many loads side by side as it takes for chunks to arrive at the rate the consumer takes them
            want = int(math.ceil(self.load_s[0] / max(self.consume_s[0], 1e-6)))
            self.depth = max(self.depth, min(self.max_depth, max(1, want)))
            self._win_start_n, self._win_start_t = 1, now
            return
        span = n + 1 - self._win_start_n                   # chunks handed over in this window
        if span < max(2, self.depth):
            return
        wall = max(now - self._win_start_t, 1e-9)
        rate = span / wall
        waited = sum(self.wait_s[self._win_start_n:n + 1]) / wall
        if self._prev_rate is not None and rate < 1.1 * self._prev_rate:
            # the last extra loader did not make the chunks arrive faster (the loader is parallel inside already): back, and stay
            self.depth = max(1, self.depth - 1)
            self._frozen = True
        elif waited > 0.1 and self.depth < self.max_depth:
            self._prev_rate = rate
            self.depth += 1
        else:
            self._prev_rate = None
        self._win_start_n, self._win_start_t = n + 1, now

    # -- consumer side ----------------------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[int, object]]:
        try:
            t_back = _time.perf_counter()
            for n in range(len(self._chunks)):
                if self._closed:
                    return
                self.depth_history.append(self.depth)
                if self.depth == 0:
                    t0 = _time.perf_counter()
                    loaded = self._timed_load(n)
                    self.wait_s[n] = _time.perf_counter() - t0
                else:
                    # chunk n itself (first pass) and the `depth` chunks after it; the loads behind n started while n - 1 was
                    # being processed
                    self._submit_up_to(n + self.depth)
                    fut: Future = self._futures.pop(n)
                    t0 = _time.perf_counter()
                    try:
                        loaded = fut.result()
                    finally:
                        self.wait_s[n] = _time.perf_counter() - t0
                self._chunks[n] = None   # the lazy chunk is not needed again; the loaded one belongs to the consumer
                t_out = _time.perf_counter()
                yield n, loaded
                del loaded
                t_back = _time.perf_counter()
                self.consume_s[n] = t_back - t_out
                self._adapt(n, t_back)
        finally:
            self.close()

    def close(self) -> None:
        """Cancel queued loads, wait for running ones, release the threads.  Idempotent."""
        self._closed = True
        pool, self._pool = self._pool, None
        for fut in self._futures.values():
            fut.cancel()
        self._futures.clear()
        if pool is not None:
            pool.shutdown(wait=True)

    @property
    def stats(self) -> dict:
        return {"depth": self.depth, "workers": self.workers, "adaptive": self.adaptive, "max_depth": self.max_depth,
                "depth_per_chunk": list(self.depth_history), "chunks": len(self._chunks), "load_s": round(sum(self.load_s), 6),
                "waited_s": round(sum(self.wait_s), 6), "consumed_s": round(sum(self.consume_s), 6),
                "load_s_per_chunk": [round(v, 6) for v in self.load_s], "waited_s_per_chunk": [round(v, 6) for v in self.wait_s]}


# the last run's statistics of get_ffpiv's executor (bench.py's `lazy_host_chunks` and the tests read it): per process, informational
LAST_STATS: dict = {}
