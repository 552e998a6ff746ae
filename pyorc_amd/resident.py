"""A LAZY frame stack kept resident in HBM while ``get_ffpiv`` runs: loads cut where dask cuts, launches cut where the kernels anchor.

The reference's chunk loop (pyorc/velocimetry/ffpiv.py:119-142, :399-440) ties three things to ONE number, ``chunksize``: how much of
the lazy stack ``.load()`` materialises on the host at a time, how much the engine computes per call, and -- through the one-frame halo
of ``frames[chunk * chunksize - 1 : (chunk + 1) * chunksize]`` (:140) -- which frames are materialised TWICE (with dask's 20-frame
blocks, pyorc/api/video.py:48,528, a whole block of decode + orthoprojection per chunk boundary).  On the MI355X the three have
different natural sizes:

* **loads** should be small enough that several exist (so that chunk n + 1 .. n + depth can be materialised while chunk n crosses
  PCIe, ``pyorc_amd.executor``), small enough that ``max_depth + 1`` of them fit the HOST budget the reference plans with
  (``available_memory() / memory_factor``, ffpiv.py:129), and cut on dask's own block boundaries so that no block is computed twice;
* **launches** must start on the time-walking kernels' anchors (``window.chunk_alignment``: every 25 or 75 pairs of the ABSOLUTE pair
  index) for the result to be the same bits whatever the chunking;
* the **frames** a launch reads only have to be in HBM, and 288 GB hold any stack pyorc meets (1 000 float32 1080p frames: 8.3 GB).

So a lazy run keeps the (narrowed) stack resident: :class:`ResidentStack` receives the loaded pieces in time order -- no halo frame is
loaded, a piece starts where the previous one ended --, uploads each to its place (``lspiv_upload_frames``: exactly the staging of the
PIV host entry points; or, for the direct product of ``project_hip``, the CAMERA frames, projected into place by the orthoprojection
kernel: the ortho frames never exist on the host), and launches every pair up to the last anchor that has both its frames.  The bits
are those of one launch over the whole stack, hence those of the chunked host path and of the reference's independent windows.

Stacks beyond the HBM budget run as consecutive windows of this kind, cut on the anchors, each re-loading one halo frame.
"""

from __future__ import annotations

import math
import threading
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .device import DeviceFrames

MIN_LOADS = 8     # a stack with at least this many anchors is cut into at least this many loads: something to run ahead of, and a short first load


def time_blocks(frames) -> Optional[List[int]]:
    """Block boundaries ``[0, b1, ..., T]`` of the time axis of a dask-backed stack (``xr.DataArray.chunks`` / ``dask.array.Array.chunks``:
    a tuple of per-axis block-length tuples; pyorc reads videos in blocks of 20 frames, pyorc/api/video.py:48,528), or None."""
    ch = getattr(frames, "chunks", None)
    if ch is None:
        data = getattr(frames, "data", None)
        ch = getattr(data, "chunks", None) if data is not None and not isinstance(data, (np.ndarray, memoryview)) else None
    try:
        if not ch or not len(ch[0]):
            return None
        lens = [int(c) for c in ch[0]]
    except (TypeError, ValueError, IndexError):
        return None
    if any(c <= 0 for c in lens) or sum(lens) != len(frames):
        return None
    return [0] + list(np.cumsum(lens))


def plan_loads(n_frames: int, load_frames: int, blocks: Optional[Sequence[int]] = None, first: int = 0) -> List[Tuple[int, int]]:
    """Cut frames ``[first, n_frames)`` into consecutive loads ``[(f0, f1), ...]`` of at most ``load_frames`` frames, on dask's block
    boundaries where there are any (a load = whole blocks, as many as fit; a block larger than the budget is split, it cannot be helped)."""
    load_frames = max(1, int(load_frames))
    if not blocks:     # nothing to align with: loads of equal length (no one-frame load at the end)
        n = max(0, int(n_frames) - int(first))
        k = max(1, -(-n // load_frames))
        edges = [int(first) + (n * i) // k for i in range(k + 1)]
        return [(a, b) for a, b in zip(edges, edges[1:]) if b > a]
    bounds = sorted({int(b) for b in (blocks or []) if first < b < n_frames} | {int(n_frames)})
    cuts = [int(first)]
    while cuts[-1] < n_frames:
        reach = cuts[-1] + load_frames
        inside = [b for b in bounds if cuts[-1] < b <= reach]
        cuts.append(inside[-1] if inside else min(reach, n_frames))
    return [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]


def load_size(n_frames: int, align: int, host_frames: int, blocks: Optional[Sequence[int]] = None) -> int:
    """Frames per load: the overlap granule -- whole anchors, so many that a stack of ``MIN_LOADS`` or more anchors gives at least
    ``MIN_LOADS`` loads --, no more than the host budget allows (``host_frames``), and whole dask blocks when that is possible."""
    n_pairs = max(1, n_frames - 1)
    n_anchors = -(-n_pairs // max(1, align))
    granule = max(1, align) * max(1, n_anchors // MIN_LOADS)
    L = max(1, min(granule, int(host_frames)))
    if blocks and len(blocks) > 2:
        typical = int(np.median(np.diff(blocks)))
        if L >= typical > 0:
            L = (L // typical) * typical
    return L


class ResidentStack:
    """Frames ``[first, first + capacity)`` of a run in HBM; pieces pushed in time order, launches on the anchors.

    ``launch(view, p0, p1)`` is called with a :class:`DeviceFrames` view of frames ``p0 .. p1`` (inclusive: ``p1 - p0`` pairs) whenever
    pairs ``[p0, p1)`` have become computable: ``p0`` is where the previous launch ended (``first`` at the start, a multiple of
    ``align`` afterwards), ``p1`` the last multiple of ``align`` below the frames that have arrived -- everything at :meth:`finish`.
    ``projection``: a ``pyorc_amd.project.Projection``; pieces are then CAMERA frames, uploaded to a scratch stack and projected into
    place (``Projection.project_into``), float32 out like ``project_hip``'s blocks."""

    def __init__(self, first: int, capacity: int, frame_shape, host_dtype, align: int, launch: Callable, signal_threshold=None,
                 projection=None):
        self.first, self.capacity = int(first), int(capacity)
        self.frame_shape = (int(frame_shape[0]), int(frame_shape[1]))
        self.align = max(1, int(align))
        self._launch = launch
        self.signal_threshold = signal_threshold
        self.projection = projection
        self.dtype = np.dtype(np.float32) if projection is not None else DeviceFrames.device_dtype(host_dtype)
        self.stack = DeviceFrames.empty((self.capacity,) + self.frame_shape, self.dtype)
        self.run_start = self.first     # first frame of the current gap-free run
        self.have = self.first          # frames [run_start, have) are resident
        self.launched = self.first      # pairs [run_start, launched) have been issued
        self.upload_s = 0.0
        self.launch_s = 0.0
        self._lock = threading.Lock()

    # pair p = frames p, p + 1 (absolute indices)
    def stage(self, f0: int, frames) -> int:
        """Bring host frames ``[f0, f0 + n)`` into their place in the stack (upload, or upload + orthoprojection); returns ``n``.  Touches
        nothing but that slice: pieces may be staged by several threads at once and in any order (the chunk executor's loaders stage
        what they loaded -- uploads queue on the library's host lock, i.e. on PCIe, while the consumer's thread only launches)."""
        import time as _time

        n = len(frames)
        if n == 0:
            return 0
        if f0 < self.first or f0 + n > self.first + self.capacity:
            raise ValueError(f"piece [{f0}, {f0 + n}) is outside [{self.first}, {self.first + self.capacity})")
        t0 = _time.perf_counter()
        if self.projection is None:
            self.stack.upload(f0 - self.first, frames, self.signal_threshold)
        else:
            cam = DeviceFrames.from_host(frames)
            self.projection.project_into(cam, self.stack, f0 - self.first)
            del cam      # stream-ordered: the block goes back to the pool, the next upload waits for the library's stream first
        with self._lock:
            self.upload_s += _time.perf_counter() - t0
        return n

    def commit(self, f0: int, n: int) -> None:
        """Frames ``[f0, f0 + n)`` have been staged: in time order, from ONE thread.  Launches what has become computable.  ``f0`` beyond
        what has arrived (a loader dropped trailing frames, ``load_frame_chunk``'s TypeError retry) closes the current run -- its last
        pairs are launched -- and starts a new one at ``f0``."""
        if n == 0:
            return
        if f0 < self.have:
            raise ValueError(f"piece [{f0}, {f0 + n}) does not follow frame {self.have}")
        if f0 > self.have:
            self._launch_ready(final=True)
            self.run_start = self.have = self.launched = f0
        self.have = f0 + n
        self._launch_ready(final=False)

    def push(self, f0: int, frames) -> None:
        """``stage`` + ``commit`` on the calling thread."""
        if f0 < self.have:
            raise ValueError(f"piece [{f0}, {f0 + len(frames)}) does not follow frame {self.have}")
        self.commit(f0, self.stage(f0, frames))

    def finish(self) -> None:
        self._launch_ready(final=True)

    def _launch_ready(self, final: bool) -> None:
        import time as _time

        last = self.have - 1                       # pairs [launched, last) have both frames
        p1 = last if final else (last // self.align) * self.align
        if p1 <= self.launched:
            return
        t0 = _time.perf_counter()
        view = self.stack[self.launched - self.first:p1 + 1 - self.first]
        self._launch(view, self.launched, p1)
        self.launch_s += _time.perf_counter() - t0
        self.launched = p1


def hbm_windows(n_frames: int, frames_per_window: int, align: int, blocks: Optional[Sequence[int]] = None) -> List[Tuple[int, int]]:
    """Frame ranges ``[(w0, w1), ...]`` (``w1`` exclusive, consecutive ranges share one frame) of the windows a run is cut into when the
    whole stack does not fit the HBM budget: ``frames_per_window`` frames each at most, window starts on multiples of ``align`` (and of
    dask's block length when that leaves at least one anchor)."""
    if n_frames <= frames_per_window:
        return [(0, n_frames)]
    step = max(1, ((frames_per_window - 1) // align) * align)           # pairs per window
    if blocks and len(blocks) > 2:
        typical = int(np.median(np.diff(blocks)))
        both = align * typical // math.gcd(align, typical) if typical > 0 else align
        if step >= both:
            step = (step // both) * both
    n_pairs = n_frames - 1
    return [(p, min(p + step, n_pairs) + 1) for p in range(0, n_pairs, step)]
