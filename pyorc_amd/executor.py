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

Memory: ``depth`` loaded chunks wait next to the one being processed (default depth 1: two chunks of frames on the host).  The
reference's planner sizes a chunk for its 4-15 x larger window stack plus the correlation volume (ffpiv.window.required_memory),
neither of which exists here, so two chunks of plain frames stay well inside that budget; ``depth = 0`` is the reference's serial loop.
"""

from __future__ import annotations

import os
import threading
import time as _time
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

DEFAULT_DEPTH = 1
DEFAULT_WORKERS = 1


def default_depth() -> int:
    """Prefetch depth when the caller does not say: ``LSPIV_PREFETCH_DEPTH`` (0 = the reference's serial loop), else 1."""
    v = os.environ.get("LSPIV_PREFETCH_DEPTH")
    if v is None or v == "":
        return DEFAULT_DEPTH
    d = int(v)
    if d < 0:
        raise ValueError(f"LSPIV_PREFETCH_DEPTH must be >= 0, got {d}")
    return d


def default_workers() -> int:
    """Loader threads when the caller does not say: ``LSPIV_PREFETCH_WORKERS``, else 1 (a dask ``.load()`` is parallel inside; more
    than one thread pays for loaders that run on one core each -- plain numpy, a single-threaded decoder -- and needs
    ``depth >= workers`` to have that many chunks in flight)."""
    v = os.environ.get("LSPIV_PREFETCH_WORKERS")
    if v is None or v == "":
        return DEFAULT_WORKERS
    w = int(v)
    if w < 1:
        raise ValueError(f"LSPIV_PREFETCH_WORKERS must be >= 1, got {w}")
    return w


class ChunkPrefetcher:
    """Iterate ``(n, load(chunks[n]))`` in order, with up to ``depth`` loads running ahead on ``workers`` threads.

    * ``chunks``: the lazy chunks (``frames[a:b]`` slices of an ``xr.DataArray``); the list is NOT modified, but the prefetcher
      drops its own reference to a chunk as soon as its load has been handed over.
    * ``load``: ``load_frame_chunk`` of ``pyorc_amd.velocimetry`` (``da.load()`` with the reference's ``TypeError`` retry).
    * ``depth``: how many chunks may be loaded (or loading) beyond the one the consumer holds.  0 = load on the caller's
      thread when the chunk is asked for: the reference's order of events.
    * ``workers``: loader threads (default 1: loads run one after the other, in chunk order, like the serial loop's -- a dask
      ``.load()`` is itself parallel inside; more than one only pays for loaders that do not use the cores themselves).
    * Errors: an exception raised by ``load(chunks[k])`` is re-raised by the iterator when chunk ``k`` is due, after chunks
      ``0 .. k-1`` have been handed over; loads queued behind it are cancelled.  Leaving the loop early (``break``, an
      exception in the consumer, ``close()``) cancels what has not started and waits for what has.

    ``stats`` after (or during) the run: per chunk the seconds its load took and the seconds the consumer waited for it; the
    wall time the consumer spent waiting is what prefetching could NOT hide.
    """

    def __init__(self, chunks: Sequence, load: Callable, depth: Optional[int] = None, workers: Optional[int] = None):
        self._chunks: List = list(chunks)
        self._load = load
        self.depth = default_depth() if depth is None else int(depth)
        if self.depth < 0:
            raise ValueError(f"prefetch depth must be >= 0, got {self.depth}")
        self.workers = default_workers() if workers is None else max(1, int(workers))
        self._pool: Optional[ThreadPoolExecutor] = None
        self._futures: dict = {}
        self._next_submit = 0
        self._lock = threading.Lock()
        self.load_s: List[float] = [0.0] * len(self._chunks)
        self.wait_s: List[float] = [0.0] * len(self._chunks)
        self._closed = False

    def __len__(self) -> int:
        return len(self._chunks)

    # -- loader side ------------------------------------------------------------------------------------------------------------
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
            self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="lspiv-load")
        last = min(last, len(self._chunks) - 1)
        while self._next_submit <= last:
            n = self._next_submit
            self._futures[n] = self._pool.submit(self._timed_load, n)
            self._next_submit += 1

    # -- consumer side ----------------------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[int, object]]:
        try:
            for n in range(len(self._chunks)):
                if self._closed:
                    return
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
                yield n, loaded
                del loaded
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
        return {"depth": self.depth, "workers": self.workers, "chunks": len(self._chunks), "load_s": round(sum(self.load_s), 6),
                "waited_s": round(sum(self.wait_s), 6), "load_s_per_chunk": [round(v, 6) for v in self.load_s],
                "waited_s_per_chunk": [round(v, 6) for v in self.wait_s]}


# the last run's statistics of get_ffpiv's executor (bench.py's `lazy_host_chunks` and the tests read it): per process, informational
LAST_STATS: dict = {}
