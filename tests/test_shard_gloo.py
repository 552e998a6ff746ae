"""World-size-2 (and 3) tests of the multi-GPU path on CPU: shard -> compute -> all-gather.

Two communicators drive the same `pyorc_amd.shard` code: the product one (`pyorc_amd.comm.Comm`, the C ABI's
lspiv_comm_* entry points -- here over their shared-memory test transport, since RCCL needs GPUs) and a
`torch.distributed` gloo adapter that lives in this file only.  The compute callable is the C oracle (tests may use
it); on the GPU box the same code path runs with the HIP engine over RCCL (bench.py --gpus N).
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

from oracle import c_oracle  # noqa: E402
from pyorc_amd import shard  # noqa: E402
from pyorc_amd.synth import particle_stack  # noqa: E402

WS, OV = (32, 32), (16, 16)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_compute(frames, ws, ov, thr, pair_offset=0):
    return c_oracle.piv_pairs(frames, ws, ov, thr, nthreads=1)


class GlooComm:
    """torch.distributed (gloo) behind the interface of pyorc_amd.comm.Comm -- test-only."""

    def __init__(self, rank, world, port):
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def allgather(self, arr):
        t = torch.as_tensor(np.ascontiguousarray(arr))
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype)
        self.dist.all_gather_into_tensor(out.view(-1), t.view(-1))
        return out.numpy()

    def allreduce(self, arr, op=shard.SUM):
        t = torch.as_tensor(np.ascontiguousarray(arr).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == shard.SUM else self.dist.ReduceOp.MAX)
        return t.numpy()

    def close(self):
        self.dist.destroy_process_group()


def _make_comm(kind, rank, world, port, out_dir):
    if kind == "gloo":
        return GlooComm(rank, world, port)
    from pyorc_amd.comm import Comm

    return Comm(rank, world, transport="shm", id_file=os.path.join(out_dir, "comm_id"), timeout=60)


def _worker(rank, world, port, n_frames, out_dir, kind, align):
    comm = _make_comm(kind, rank, world, port, out_dir)
    try:
        stack = particle_stack(n_frames, 96, 128, seed=42)
        touched = []

        def load(a, b):
            touched.append((a, b))
            return stack[a:b]

        full = shard.sharded_piv(load, n_frames - 1, WS, OV, comm, compute=_oracle_compute, align=align)
        summed = comm.allreduce(np.full((3, 2), rank + 1.0, np.float32), shard.SUM)
        biggest = comm.allreduce(np.array([rank * 1.5], np.float64), shard.MAX)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full, touched=np.array(touched), summed=summed, biggest=biggest)
    finally:
        comm.close()


@pytest.mark.parametrize("kind", ["native-shm", "gloo"])
@pytest.mark.parametrize("world,n_frames,align", [(2, 7, 1), (2, 6, 1), (3, 3, 1), (2, 8, 3), (3, 9, 5)])
def test_sharded_piv_equals_single_process(tmp_path, world, n_frames, align, kind):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path), kind, align), nprocs=world, join=True)
    stack = particle_stack(n_frames, 96, 128, seed=42)
    ref = np.stack(_oracle_compute(stack, WS, OV, None))
    n_pairs = n_frames - 1
    _check_ranks(tmp_path, world, n_frames, align, ref)


def _check_ranks(tmp_path, world, n_frames, align, ref):
    n_pairs = n_frames - 1
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        assert d["full"].shape == ref.shape
        assert np.array_equal(d["full"], ref, equal_nan=True)  # bit-identical on every rank
        a, b = shard.frame_block(n_pairs, r, world, align)
        assert a % align == 0  # every block starts on an anchor of the time-walking kernels
        if b - a >= 2:
            assert d["touched"].tolist() == [[a, b]]  # a rank reads only its block + one halo frame
        else:
            assert d["touched"].size == 0
        assert np.all(d["summed"] == sum(range(1, world + 1)))
        assert d["biggest"][0] == (world - 1) * 1.5


@pytest.mark.parametrize("world,n_frames,align", [(4, 14, 3), (8, 27, 3), (8, 6, 1)])
def test_sharded_piv_at_the_rank_counts_of_the_scaling_run(tmp_path, world, n_frames, align):
    """4 and 8 ranks (the driver's N = 4, 8) over the native communicator's shared-memory transport, with more ranks than
    anchored blocks in the last case (ranks with nothing to do still take part in the gather)."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path), "native-shm", align), nprocs=world, join=True)
    stack = particle_stack(n_frames, 96, 128, seed=42)
    _check_ranks(tmp_path, world, n_frames, align, np.stack(_oracle_compute(stack, WS, OV, None)))


class OracleEnsemble:
    """numpy stand-in with the interface of pyorc_amd.piv.Ensemble (accumulate / export / import / finish), built on
    the oracle, so that the sharded-ensemble plumbing can run on CPU ranks."""

    def __init__(self, dim_size, ws, ov):
        from oracle import piv_oracle as po

        self.po, self.ws, self.ov = po, ws, ov
        x, y = po.get_rect_coordinates(dim_size, ws, ov)
        self.n_rows, self.n_cols = len(y), len(x)
        n = self.n_rows * self.n_cols
        self.s = np.zeros((n,) + ws, np.float64)
        self.k = np.zeros(n, np.float64)

    def accumulate(self, frames, corr_min, s2n_min, thr=None, out=None):
        _, _, corr = self.po.cross_corr(frames, self.ws, self.ov, signal_threshold=thr)
        with np.errstate(all="ignore"):
            cm = corr.max(axis=(-1, -2))
            sn = cm / corr.mean(axis=(-1, -2))
        keep = (cm >= corr_min) & (sn >= s2n_min) & np.isfinite(cm)
        corr[~keep] = 0.0; cm[~keep] = 0.0; sn[~keep] = 0.0
        self.s += corr.sum(axis=0)
        self.k += (cm > 1e-6).sum(axis=0)
        if out is not None:     # pyorc_amd.piv.Ensemble.accumulate(out=): the caller's time slices receive the results
            out[0][...] = cm
            out[1][...] = sn
            return out
        return cm.astype(np.float32), sn.astype(np.float32)

    def export_state(self):
        return self.s.astype(np.float32), self.k.astype(np.float32)

    def import_state(self, s, k, add=False):
        self.s = np.asarray(s, np.float64).reshape(self.s.shape) + (self.s if add else 0)
        self.k = np.asarray(k, np.float64) + (self.k if add else 0)

    # the staged finish of pyorc_amd.piv.Ensemble (flag / partials / finish_partials): the float64 oracle never has an ill-conditioned
    # float32 fit to re-evaluate, so it flags `n_flag` windows only to drive the exchange (a rank's partials are its rank + 1)
    n_flag = 0

    def flag(self, count_min, n_frames):
        self._args = (count_min, n_frames)
        return self.n_flag

    def flag_digest(self):
        return getattr(self, "digest", 0xABCDEF0123456789)

    def partials(self):
        return np.full((self.n_flag, 20), 1.0 + getattr(self, "rank", 0), np.float64), not getattr(self, "lost_frames", False)

    def finish_partials(self, partials):
        self.summed_partials = np.array(partials)
        return self.finish(*self._args)

    def finish(self, count_min, n_frames):
        self.plain_finish = not hasattr(self, "summed_partials")
        with np.errstate(all="ignore"):
            mean = self.s / self.k[:, None, None]
            mean[self.k < count_min * n_frames] = np.nan
        u, v = self.po.u_v_displacement(mean[None], self.n_rows, self.n_cols)
        return u.astype(np.float32), v.astype(np.float32), self.k.astype(np.float32)


def _ens_worker(rank, world, port, n_frames, out_dir, kind):
    comm = _make_comm(kind, rank, world, port, out_dir)
    try:
        stack = particle_stack(n_frames, 64, 96, seed=43)
        made = []

        def make():
            e = OracleEnsemble((64, 96), WS, OV)
            e.rank, e.n_flag = rank, 3                      # three "flagged" windows: the float64 all-reduce of the partials runs
            e.lost_frames = (n_frames == 5 and rank == 1)    # one rank of the (3, 5) case could not keep its frames: all ranks fall back
            if n_frames == 9:                                # the (2, 9) case: as many flagged windows everywhere, but not the same ones (digests differ in the low / the high half)
                e.digest = 0xABCDEF0123456789 + (rank if kind == "gloo" else rank << 40)
            made.append(e)
            return e

        u, v, cnt, cm, sn = shard.sharded_ensemble(lambda a, b: stack[a:b], n_frames - 1, make, 0.2, 2.0, 0.2, comm, n_chunks=2)
        e = made[0]
        np.savez(os.path.join(out_dir, f"e{rank}.npz"), u=u, v=v, cnt=cnt, cm=cm, sn=sn, plain=e.plain_finish,
                 summed=getattr(e, "summed_partials", np.zeros((0, 20))))
    finally:
        comm.close()


@pytest.mark.parametrize("kind", ["native-shm", "gloo"])
@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 5), (2, 9)])
def test_sharded_ensemble_equals_single_process(tmp_path, world, n_frames, kind):
    port = _free_port()
    mp.spawn(_ens_worker, args=(world, port, n_frames, str(tmp_path), kind), nprocs=world, join=True)
    stack = particle_stack(n_frames, 64, 96, seed=43)
    ref = OracleEnsemble((64, 96), WS, OV)
    cm, sn = ref.accumulate(stack, 0.2, 2.0)
    u, v, cnt = ref.finish(0.2, 2)   # n_chunks is an argument: the count filter does not depend on the world size
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"e{r}.npz"))
        assert np.array_equal(d["cnt"], cnt) and np.array_equal(d["cm"], cm) and np.array_equal(d["sn"], sn)
        assert np.array_equal(np.isnan(d["u"]), np.isnan(u))
        # the per-rank partial sums are added in float32 by the all-reduce: same peak, sub-pixel within 1e-4
        assert np.nanmax(np.abs(d["u"] - u)) < 1e-4 and np.nanmax(np.abs(d["v"] - v)) < 1e-4
        # the staged finish: every rank's partials summed over the ranks (1 + 2 [+ 3]); if any rank lost its frames, all finish plainly
        if n_frames in (5, 9):    # ... and so they do when their lists of flagged windows differ (same length, other digest)
            assert bool(d["plain"]) and d["summed"].size == 0
        else:
            assert not bool(d["plain"]) and d["summed"].shape == (3, 20) and np.all(d["summed"] == sum(range(1, world + 1)))
