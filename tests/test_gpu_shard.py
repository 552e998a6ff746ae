"""The product's own sharding API with the HIP engine on a GPU: `world` ranks on device 0 over the shared-
memory transport of the C ABI (RCCL refuses two ranks on one GPU; the RCCL transport runs with one rank in test_gpu_bench.py),
``shard.sharded_piv`` with its DEFAULT compute and the walking kernels' alignment of 25 pairs -- rank blocks start at non-zero
anchors -- against ONE ``piv.piv_pairs`` launch over the whole stack, bit for bit; the device-resident variant
(``ShardedPivDev`` / ``sharded_piv_dev``, what bench.py --gpus N times); and ``sharded_ensemble`` with ``piv.Ensemble``."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pyorc_amd import shard
from pyorc_amd.synth import particle_stack

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WS, OV = (32, 32), (16, 16)


def run_ranks(mode, world, n_frames, out_dir):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LSPIV_COMM_NONCE=f"t{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_gpu_worker.py"), mode, str(out_dir), str(n_frames)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(out_dir, f"r{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("world,n_frames", [(2, 86), (3, 86), (3, 41)])
def test_sharded_piv_default_compute_equals_one_launch(gpu, tmp_path, world, n_frames):
    import pyorc_amd

    res = run_ranks("piv", world, n_frames, tmp_path)
    stack = particle_stack(n_frames, 96, 128, seed=77, density=0.03)
    ref = np.stack(pyorc_amd.piv_pairs(stack, WS, OV))
    n_pairs = n_frames - 1
    starts = []
    for r, d in enumerate(res):
        assert int(d["align"]) == 25
        assert d["full"].shape == ref.shape and np.array_equal(d["full"], ref, equal_nan=True), r     # the single-GPU bits on every rank
        a, b = shard.frame_block(n_pairs, r, world, 25)
        assert d["touched"].tolist() == ([[a, b]] if b - a >= 2 else [])                              # only its block + the halo frame
        starts.append(a)
    assert any(a > 0 and a % 25 == 0 for a in starts)                                                 # a walking-kernel launch with pair_offset != 0


@pytest.mark.parametrize("world,n_frames", [(2, 86), (3, 61)])
def test_sharded_piv_dev_equals_one_launch(gpu, tmp_path, world, n_frames):
    import pyorc_amd

    res = run_ranks("piv_dev", world, n_frames, tmp_path)
    stack = particle_stack(n_frames, 96, 128, seed=77, density=0.03)
    ref = np.stack(pyorc_amd.piv_pairs(stack, WS, OV))
    for r, d in enumerate(res):
        assert np.array_equal(d["full"], ref, equal_nan=True) and np.array_equal(d["one"], ref, equal_nan=True), r
        assert d["block"].tolist() == list(shard.frame_block(n_frames - 1, r, world, 25))
        assert d["kernel_ms"].shape == (3,) and np.all(d["kernel_ms"] > 0) and np.all(d["gather_ms"] > 0)


@pytest.mark.parametrize("world,n_frames", [(2, 12), (3, 80)])
def test_sharded_ensemble_with_the_hip_engine(gpu, tmp_path, world, n_frames):
    """Every rank accumulates its block into its own ``piv.Ensemble``; sums all-reduced, every rank finishes -- incl. the float64
    rescue of ill-conditioned fits through the staged finish (each rank contributes the sums over its own frames)."""
    from oracle import piv_oracle as po

    res = run_ranks("ensemble", world, n_frames, tmp_path)
    stack = particle_stack(n_frames, 96, 128, seed=77, density=0.03)
    ref = po.get_ffpiv(stack, np.ones(n_frames - 1), WS, OV, 1.0, 1.0, ensemble_corr=True, corr_min=0.1, s2n_min=1.5, count_min=0.2)
    uo, vo = ref["v_x"][0].astype(np.float64), ref["v_y"][0].astype(np.float64)
    for r, d in enumerate(res):
        u, v = d["u"][0].astype(np.float64), d["v"][0].astype(np.float64)
        assert np.array_equal(np.isnan(u), np.isnan(uo)), r
        e = max(np.nanmax(np.abs(u - uo) / np.maximum(np.abs(uo), 0.05)), np.nanmax(np.abs(v - vo) / np.maximum(np.abs(vo), 0.05)))
        assert e <= 1e-4, (r, e)
        assert np.array_equal(d["u"], res[0]["u"], equal_nan=True) and np.array_equal(d["cnt"], res[0]["cnt"])   # all ranks agree
        assert d["cm"].shape == (n_frames - 1, uo.size)
