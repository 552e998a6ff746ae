"""Frame-pair sharding over the GPUs of one node (one process per GPU).

PIV pairs are independent (docs/user-guide/velocimetry/index.rst:12-13 of the reference), and the
reference already splits the time axis into chunks with a 1-frame halo
(pyorc/velocimetry/ffpiv.py:140).  Sharding is the same cut, one level up: rank r of R owns the
contiguous pair block [r P / R, (r+1) P / R) and therefore reads frames [r P / R, (r+1) P / R] --
the halo frame is read redundantly, never communicated.  The only exchange is ONE all-gather of each
rank's packed (4, p_local, n_rows, n_cols) float32 result block (u | v | corr_max | s2n) at the end;
ensemble mode instead needs a sum all-reduce of corr_sum / corr_count.

The communicator is ``pyorc_amd.comm.Comm`` (RCCL over xGMI through the C ABI, no PyTorch); anything with
``rank``, ``world``, ``allgather(arr)`` and ``allreduce(arr, op)`` works (the CPU tests also drive this module
through a ``torch.distributed`` gloo adapter).  The compute callable is injected.

Block boundaries are multiples of ``align`` pairs (``window.chunk_alignment``), so every rank starts on an anchor of
the time-walking kernels' segments: the gathered result equals the single-GPU result bit for bit.
"""

from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import numpy as np

SUM, MAX = 0, 1


def pair_block(n_pairs: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """[start, stop) of the pairs owned by ``rank``; blocks are contiguous and ordered; their sizes differ by at most
    one ``align``-sized unit (boundaries are multiples of ``align``, the tail goes to the last non-empty block)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if align <= 1:
        return (n_pairs * rank) // world, (n_pairs * (rank + 1)) // world
    units = -(-n_pairs // align)  # ceil: the last unit may be short
    a = min(((units * rank) // world) * align, n_pairs)
    b = min(((units * (rank + 1)) // world) * align, n_pairs)
    return a, b


def frame_block(n_pairs: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """[start, stop) of the FRAMES rank needs: its pairs plus the one-frame halo."""
    a, b = pair_block(n_pairs, rank, world, align)
    return (a, b + 1) if b > a else (a, a)


def block_sizes(n_pairs: int, world: int, align: int = 1) -> List[int]:
    return [pair_block(n_pairs, r, world, align)[1] - pair_block(n_pairs, r, world, align)[0] for r in range(world)]


def gather_blocks(local: np.ndarray, n_pairs: int, comm, align: int = 1) -> np.ndarray:
    """All-gather per-rank result blocks (k, p_local, n_rows, n_cols) into (k, n_pairs, n_rows, n_cols).

    Blocks may differ in length; they are padded to the largest block so that ONE all-gather moves everything
    (one large collective, not one per variable).
    """
    sizes = block_sizes(n_pairs, comm.world, align)
    pmax = max(sizes)
    k, p_local, n_rows, n_cols = local.shape
    if p_local != sizes[comm.rank]:
        raise ValueError(f"rank {comm.rank} holds {p_local} pairs, expected {sizes[comm.rank]}")
    send = np.full((k, pmax, n_rows, n_cols), np.nan, dtype=np.float32)
    send[:, :p_local] = local
    recv = comm.allgather(send)
    return np.concatenate([recv[r][:, :sizes[r]] for r in range(comm.world)], axis=1)


def sharded_piv(load_frames: Callable[[int, int], np.ndarray], n_pairs: int, window_size, overlap, comm,
                compute: Optional[Callable] = None, signal_threshold=None, align: Optional[int] = None,
                frame_shape=None) -> np.ndarray:
    """Every rank computes its pair block and all ranks receive the full (4, n_pairs, n_rows, n_cols) block.

    ``load_frames(start, stop)`` returns frames [start, stop) as (T, H, W) -- each rank only ever touches its
    own time block (+ halo).  ``compute(frames, window_size, overlap, signal_threshold, pair_offset=...)`` ->
    (u, v, corr_max, s2n); default is the HIP engine (``pyorc_amd.piv.piv_pairs``).
    ``frame_shape`` (H, W): the rank blocks are cut on the anchor length of THAT window grid (``window.chunk_alignment``: 25 pairs, 75 on
    large grids); without it on the longest anchor of the window family, which is right for every grid.
    """
    if compute is None:
        from . import piv

        compute = piv.piv_pairs
    if align is None:
        from . import window

        align = window.chunk_alignment(window_size, frame_shape, overlap) if frame_shape is not None else window.chunk_alignment_any_grid(window_size)
    f0, f1 = frame_block(n_pairs, comm.rank, comm.world, align)
    if f1 - f0 >= 2:
        u, v, cm, sn = compute(load_frames(f0, f1), window_size, overlap, signal_threshold, pair_offset=f0)
        local = np.stack([u, v, cm, sn]).astype(np.float32)
        shape = np.array(local.shape[2:], dtype=np.float64)
    else:
        local, shape = None, np.zeros(2, dtype=np.float64)
    # ranks without pairs (n_pairs < world) learn the grid shape from the others
    shape = comm.allreduce(shape, MAX)
    if local is None:
        local = np.empty((4, 0, int(shape[0]), int(shape[1])), dtype=np.float32)
    return gather_blocks(local, n_pairs, comm, align)


def sharded_ensemble(load_frames: Callable[[int, int], np.ndarray], n_pairs: int, make_ensemble: Callable[[], object],
                     corr_min: float, s2n_min: float, count_min: float, comm, signal_threshold=None, n_chunks: int = 1,
                     align: int = 1):
    """Ensemble correlation over the GPUs of a node (pyorc/velocimetry/ffpiv.py:182-376 sharded in time).

    Every rank accumulates its own pair block into its own ensemble object (``make_ensemble()`` -> an object with
    ``accumulate / export_state / import_state / finish``, e.g. ``pyorc_amd.piv.Ensemble``); corr_sum and corr_count
    are summed with ONE all-reduce each; every rank imports the total and finishes, so all ranks return the same
    (u, v, corr_count, corr_max (P, n_win), s2n (P, n_win)).

    ``n_chunks`` is the ``n_frames`` of the count filter ``corr_count < count_min * n_frames`` -- the reference counts
    CHUNKS there (quirk Q3, ffpiv.py:373,280-281).  It is an argument (what the single-process chunk planner yields,
    ``len(velocimetry.plan_chunks(...)[1])``; 1 for a stack that fits one chunk), NOT the number of ranks: which
    windows survive the filter must not depend on how many GPUs shared the work.

    Reproducibility: a rank's partial sum is bit-reproducible (anchored segments, fixed merge order), but the sum over
    ranks is a floating-point all-reduce and every rank's handle starts its segment anchors at its own pair 0 -- so the
    sharded mean planes agree with a single-GPU run to float32 rounding (1e-6 of the plane), not bit for bit.  The
    per-timestep path (``sharded_piv``) IS bit-identical to one GPU.  Windows whose final fit is ill-conditioned are
    re-evaluated in float64 from the frames on every rank's block (round 4), like on one GPU.
    """
    ens = make_ensemble()
    f0, f1 = frame_block(n_pairs, comm.rank, comm.world, align)
    if f1 - f0 >= 2:
        cm, sn = ens.accumulate(load_frames(f0, f1), corr_min, s2n_min, signal_threshold)
    else:
        cm = sn = None
    s, k = ens.export_state()
    s = comm.allreduce(s, SUM)
    k = comm.allreduce(k, SUM)
    ens.import_state(s, k)
    # every rank now holds the same sums.  The float64 rescue of the ill-conditioned fits needs the frames of every pair: each
    # rank contributes the float64 sums over ITS retained block for the (identical, sorted) list of flagged windows -- one more
    # float64 all-reduce of a few hundred bytes (include/lspiv.h, lspiv_ensemble_flag / _partials / _finish_partials)
    if hasattr(ens, "flag"):
        n_rec = ens.flag(count_min, n_chunks)
        part, ok = ens.partials()
        # one small MAX all-reduce settles both questions for everybody: did every rank keep its frames, and do all ranks hold the
        # same list (they do -- identical sums after the all-reduce --; a rank that disagreed would make the next exchange hang)
        # -- the same LIST, not only the same length: the partials are summed positionally, so the digest of the sorted records
        # (window, candidates) rides along as two exact 32-bit halves, each with its negative (MAX of x and of -x agree <=> all equal)
        dg = ens.flag_digest() if hasattr(ens, "flag_digest") else 0
        hi, lo = float(dg >> 32), float(dg & 0xFFFFFFFF)
        agreed = comm.allreduce(np.array([0.0 if ok else 1.0, float(n_rec), -float(n_rec), hi, -hi, lo, -lo], dtype=np.float64), MAX)
        all_ok = agreed[0] == 0.0 and agreed[1] == -agreed[2] and agreed[3] == -agreed[4] and agreed[5] == -agreed[6]
        if all_ok:
            if n_rec:
                part = comm.allreduce(part, SUM)
            u, v, cnt = ens.finish_partials(part)
        else:      # some rank could not keep its frames (or the lists differ): float32 fits everywhere (the ranks agree)
            u, v, cnt = ens.finish(count_min, n_chunks)
    else:
        u, v, cnt = ens.finish(count_min, n_chunks)
    n_win = k.size
    local = np.stack([cm, sn]).astype(np.float32)[:, :, None, :] if cm is not None else np.empty((2, 0, 1, n_win), np.float32)
    per_pair = gather_blocks(local, n_pairs, comm, align)  # (2, n_pairs, 1, n_win)
    return u, v, cnt, per_pair[0, :, 0], per_pair[1, :, 0]


class ShardedPivDev:
    """The sharded per-timestep path with everything resident in HBM: what ``sharded_piv`` does with host arrays, for a
    rank whose time block already sits on its GPU (``DeviceFrames``) -- and what ``bench.py --gpus N`` times.

    A plan for repeated use: two packed result blocks and two gather buffers per rank, a compute stream and a gather
    stream.  ``step(block)`` launches the PIV kernels of this rank's pairs (anchored at their absolute index, so the gathered
    bits equal one launch over the whole stack) and queues the all-gather of the packed ``[u | v | corr | s2n]`` block behind
    it on the gather stream; the NEXT step's kernels overlap with that gather (a buffer is reused two steps later, after its
    gather).  ``drain()`` waits for both streams; ``gathered_host()`` returns the last step's (4, n_pairs, n_rows, n_cols).
    Blocks of unequal length are padded to the longest in the exchange only (the all-gather moves equal counts).
    """

    def __init__(self, comm, n_pairs: int, frame_shape, window_size, overlap, signal_threshold=None, align: Optional[int] = None,
                 record_timings: bool = False):
        import ctypes as C

        from . import _lib, window
        from .device import DeviceFrames

        self._C, self._lib_mod, self.lib = C, _lib, _lib.load()
        self.comm, self.n_pairs = comm, int(n_pairs)
        self.window_size, self.overlap = tuple(window_size), tuple(overlap)
        self.H, self.W = int(frame_shape[0]), int(frame_shape[1])
        self.align = window.chunk_alignment(self.window_size, (self.H, self.W), self.overlap) if align is None else int(align)
        self.sizes = block_sizes(self.n_pairs, comm.world, self.align)
        self.a, self.b = pair_block(self.n_pairs, comm.rank, comm.world, self.align)
        self.p_local, self.p_max = self.b - self.a, max(self.sizes)
        self.n_rows, self.n_cols = window.get_array_shape((self.H, self.W), self.window_size, self.overlap)
        self.n_win = self.n_rows * self.n_cols
        self.thr = -1.0 if signal_threshold is None else float(signal_threshold)
        self.count = 4 * self.p_max * self.n_win                     # floats every rank sends
        self.send = [DeviceFrames.empty((4, self.p_max, self.n_win), np.float32) for _ in range(2)]
        self.recv = [DeviceFrames.empty((comm.world * 4, self.p_max, self.n_win), np.float32) for _ in range(2)]
        self.comp, self.comm_s = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.lspiv_stream_create(C.byref(self.comp)))
        # the exchange on a HIGH-priority stream: the collective's few workgroups get the next free compute units while the PIV
        # kernel of the following step (tens of thousands of workgroups) drains through all of them (LSPIV_GATHER_STREAM_PRIORITY=0: equal)
        self.gather_priority = int(os.environ.get("LSPIV_GATHER_STREAM_PRIORITY", "1"))
        _lib.check(self.lib.lspiv_stream_create_priority(C.byref(self.comm_s), self.gather_priority))
        self.ev_done = [self._event() for _ in range(2)]             # kernels of buffer b finished
        self.ev_gathered = [self._event() for _ in range(2)]         # gather of buffer b finished
        self._gathered_once = [False, False]
        self.k = 0
        self.record = bool(record_timings)
        self._marks = []                                             # per step: (kernel start, kernel stop, gather start, gather stop)

    def _event(self):
        e = self._C.c_void_p()
        self._lib_mod.check(self.lib.lspiv_event_create(self._C.byref(e)))
        return e

    def frame_block(self) -> Tuple[int, int]:
        """[start, stop) of the frames this rank needs (its pairs + the halo frame)."""
        return (self.a, self.b + 1) if self.b > self.a else (self.a, self.a)

    def step(self, block) -> None:
        """``block``: DeviceFrames with frames ``frame_block()`` of the stack (this rank's pairs + one halo frame)."""
        C, _lib, lib = self._C, self._lib_mod, self.lib
        if self.p_local and tuple(block.shape) != (self.p_local + 1, self.H, self.W):
            raise ValueError(f"rank {self.comm.rank} expects a block of shape {(self.p_local + 1, self.H, self.W)}, got {block.shape}")
        b = self.k & 1
        if self._gathered_once[b]:
            _lib.check(lib.lspiv_stream_wait_event(self.comp, self.ev_gathered[b]))      # gather k-2 has read send[b]
        marks = [self._event() for _ in range(4)] if self.record else None
        if marks:
            _lib.check(lib.lspiv_event_record_on(marks[0], self.comp))
        if self.p_local:
            _lib.check(lib.lspiv_piv_pairs_dev_at(block.c_ptr, block.dtype_code, self.p_local + 1, self.H, self.W, self.window_size[0],
                                                  self.window_size[1], self.overlap[0], self.overlap[1], self.thr, self.a,
                                                  self.send[b].c_ptr, None, self.comp))
        if marks:
            _lib.check(lib.lspiv_event_record_on(marks[1], self.comp))
        _lib.check(lib.lspiv_event_record_on(self.ev_done[b], self.comp))
        _lib.check(lib.lspiv_stream_wait_event(self.comm_s, self.ev_done[b]))
        if marks:
            _lib.check(lib.lspiv_event_record_on(marks[2], self.comm_s))
        self.comm.allgather_dev(self.send[b].ptr, self.recv[b].ptr, self.count, np.float32, self.comm_s.value)
        if marks:
            _lib.check(lib.lspiv_event_record_on(marks[3], self.comm_s))
            self._marks.append(marks)
        _lib.check(lib.lspiv_event_record_on(self.ev_gathered[b], self.comm_s))
        self._gathered_once[b] = True
        self.k += 1

    def drain(self) -> None:
        self._lib_mod.check(self.lib.lspiv_stream_synchronize(self.comp))
        self._lib_mod.check(self.lib.lspiv_stream_synchronize(self.comm_s))

    def timings(self, reset: bool = True) -> dict:
        """Per recorded step, by HIP events (after ``drain``): kernel ms (PIV + rescue kernels of this rank's block -- while the
        previous step's gather is in flight), gather ms, and how long after its kernels a step's gather ended."""
        C, _lib, lib = self._C, self._lib_mod, self.lib
        self.drain()

        def ms(e0, e1):
            v = C.c_float()
            _lib.check(lib.lspiv_event_elapsed_ms(e0, e1, C.byref(v)))
            return v.value

        out = {"kernel_ms": [ms(m[0], m[1]) for m in self._marks], "gather_ms": [ms(m[2], m[3]) for m in self._marks],
               "gather_end_after_kernel_end_ms": [ms(m[1], m[3]) for m in self._marks]}
        if reset:
            for m in self._marks:
                for e in m:
                    lib.lspiv_event_destroy(e)
            self._marks = []
        return out

    def gathered_host(self) -> np.ndarray:
        """The last step's results of ALL ranks, (4, n_pairs, n_rows, n_cols) float32 (after ``drain``)."""
        if self.k == 0:
            raise RuntimeError("no step has run")
        self.drain()
        raw = self.recv[(self.k - 1) & 1].to_host().reshape(self.comm.world, 4 * self.p_max * self.n_win)
        parts = [raw[r][: 4 * self.sizes[r] * self.n_win].reshape(4, self.sizes[r], self.n_rows, self.n_cols) for r in range(self.comm.world)]
        return np.concatenate(parts, axis=1)

    def local_host(self) -> np.ndarray:
        """This rank's own packed block of the last step, (4, p_local, n_rows, n_cols)."""
        self.drain()
        raw = self.send[(self.k - 1) & 1].to_host().reshape(-1)
        return raw[: 4 * self.p_local * self.n_win].reshape(4, self.p_local, self.n_rows, self.n_cols)

    def close(self) -> None:
        if self.comp:
            self.drain()
            for e in self.ev_done + self.ev_gathered + [e for m in self._marks for e in m]:
                self.lib.lspiv_event_destroy(e)
            self.lib.lspiv_stream_destroy(self.comp)
            self.lib.lspiv_stream_destroy(self.comm_s)
            self.comp = self.comm_s = None
            self.send = self.recv = []


def sharded_piv_dev(block, n_pairs: int, window_size, overlap, comm, signal_threshold=None, align: Optional[int] = None) -> np.ndarray:
    """One sharded pass with the rank's frames already in HBM: ``block`` = DeviceFrames of this rank's ``frame_block`` (its
    pairs + the halo frame; a rank without pairs passes an empty ``(0, H, W)`` stack -- the frame shape is still needed for the
    exchange --, ``None`` raises ValueError).  Returns (4, n_pairs, n_rows, n_cols) on every rank --
    bit-identical to ``piv.piv_pairs`` over the whole stack on one GPU."""
    if block is None:
        raise ValueError("sharded_piv_dev needs this rank's DeviceFrames block (ranks without pairs pass an empty (0, H, W) stack)")
    plan = ShardedPivDev(comm, n_pairs, block.shape[1:], window_size, overlap, signal_threshold, align)
    try:
        plan.step(block)
        return plan.gathered_host()
    finally:
        plan.close()
