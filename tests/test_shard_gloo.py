"""World-size-2 (and 3) gloo tests of the multi-GPU path on CPU: shard -> compute -> all-gather.

The compute callable is the C oracle here (tests may use it); on the GPU box the same code path runs with the
HIP engine and backend "nccl" (bench.py --gpus N).
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


def _oracle_compute(frames, ws, ov, thr):
    return c_oracle.piv_pairs(frames, ws, ov, thr, nthreads=1)


def _worker(rank, world, port, n_frames, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stack = particle_stack(n_frames, 96, 128, seed=42)
        touched = []

        def load(a, b):
            touched.append((a, b))
            return stack[a:b]

        full = shard.sharded_piv(load, n_frames - 1, WS, OV, compute=_oracle_compute)
        summed = shard.allreduce_sum(np.full((3, 2), rank + 1.0, np.float32))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full, touched=np.array(touched), summed=summed)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (2, 6), (3, 3)])
def test_sharded_piv_equals_single_process(tmp_path, world, n_frames):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    stack = particle_stack(n_frames, 96, 128, seed=42)
    ref = np.stack(_oracle_compute(stack, WS, OV, None))
    n_pairs = n_frames - 1
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        assert d["full"].shape == ref.shape
        assert np.array_equal(d["full"], ref, equal_nan=True)  # bit-identical on every rank
        a, b = shard.frame_block(n_pairs, r, world)
        if b - a >= 2:
            assert d["touched"].tolist() == [[a, b]]  # a rank reads only its block + one halo frame
        else:
            assert d["touched"].size == 0
        assert np.all(d["summed"] == sum(range(1, world + 1)))
