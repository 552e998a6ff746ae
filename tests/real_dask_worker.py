"""Round-6 host code against a REAL dask (run by tests/test_real_dask.py under /opt/conda/bin/python3.9, the one interpreter of the build
image that has dask; xarray exists nowhere here, so the DataArray around the dask array is a ten-line wrapper).

What is real: ``dask.array`` graphs, names, ``HighLevelGraph.dependencies``, chunks, the threaded scheduler running
``pyorc_amd.plugin._project_block`` on its worker threads.  What is a double: the GPU (the C oracle computes the PIV, the numpy oracle the
projection, the resident stack lives in host memory) -- the subject here is the host logic around the kernels."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSPIV_NO_AUTO_INSTALL"] = "1"

import dask  # noqa: E402
import dask.array as da  # noqa: E402

from pyorc_amd import _lib, executor, frames as F, plugin, velocimetry as V  # noqa: E402
from pyorc_amd.synth import particle_stack, projection_maps  # noqa: E402
from tests import doubles  # noqa: E402
from tests.test_round6_host import OraclePlan  # noqa: E402


class Lazy:
    """The few members of a dask-backed ``xr.DataArray`` that get_ffpiv touches."""

    def __init__(self, arr):
        self.data, self.shape, self.dtype, self.chunks = arr, arr.shape, arr.dtype, arr.chunks

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        return Lazy(self.data[key]) if isinstance(key, slice) else self.data[key]

    def load(self):
        return Loaded(self.data.compute(scheduler="threads", num_workers=4))


class Loaded:
    def __init__(self, values):
        self.values = values

    def __len__(self):
        return len(self.values)


class Setattr:
    def __init__(self):
        self.saved = []

    def setattr(self, obj, name, value):
        self.saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def undo(self):
        for obj, name, value in reversed(self.saved):
            setattr(obj, name, value)


def main():
    mp = Setattr()
    mp.setattr(V.piv, "piv_pairs", doubles.oracle_piv_pairs)
    mp.setattr(V.window, "available_memory", lambda: 1e12)
    mp.setattr(V.window, "chunk_alignment", lambda ws, dim=None, ov=None: 10)
    mp.setattr(_lib, "require_device", lambda: None)
    from pyorc_amd import project as P

    mp.setattr(P, "Projection", OraclePlan)
    stacks = doubles.use_host_stacks(mp)
    src, dst = (96, 128), (72, 100)
    maps = projection_maps(src, dst, tilt=0.2, seed=4)
    cam = particle_stack(47, src[0], src[1], seed=12)
    t = np.arange(47) / 30.0
    decoded, lock, threads = {}, threading.Lock(), set()

    def decode(block, block_info=None):
        if block_info is None or block.size == 0:
            return block
        k = block_info[0]["chunk-location"][0]
        with lock:
            decoded[k] = decoded.get(k, 0) + 1
            threads.add(threading.current_thread().name)
        return block

    video = da.from_array(cam, chunks=(10,) + src).map_blocks(decode, dtype=np.uint8)      # pyorc/api/video.py:528: blocks of frames
    assert video.chunks[0] == (10, 10, 10, 10, 7)
    plan_args = tuple(maps)
    # xr.apply_ufunc(..., dask="parallelized") hands exactly this to dask (xarray/core/computation.py: apply_gufunc with the core
    # dimensions as signature); the node that comes back is what project_hip registers
    ortho = da.apply_gufunc(plugin._project_block, "(y,x)->(ny,nx)", video, output_dtypes=np.float32, output_sizes={"ny": dst[0], "nx": dst[1]},
                            plan_args=plan_args, dst_shape=dst, device=None)
    assert ortho.shape == (47,) + dst and ortho.chunks[0] == (10, 10, 10, 10, 7)
    camera = Lazy(video)
    plugin._register_projection(Lazy(ortho), camera, plan_args, dst, None)
    filled = da.where(~da.isnan(ortho), ortho, 0.0)                                      # Frames.project's fillna(0.0) on a dask array
    assert filled.dtype == np.float32
    kw = dict(time=t, resolution=0.01)

    # (1) the direct product of project_hip: recognised on the real graph, camera blocks loaded once, nothing projected on the host
    hit = plugin.hip_projection_source(Lazy(filled))
    assert hit is not None and hit["source"] is camera, "fillna(project_hip(...)) was not recognised on a real dask graph"
    assert plugin.hip_projection_source(Lazy(ortho))["source"] is camera
    OraclePlan.made.clear()
    got = F.get_piv(Lazy(filled), 32, **kw)
    st = dict(executor.LAST_STATS)
    assert st["plan"]["source"] == "camera" and st["plan"]["load_frames"] == 10 and st["chunks"] == 5, st
    assert decoded == {k: 1 for k in range(5)}, decoded
    plan = OraclePlan.made[0]
    assert plan.blocks == [] and sorted(plan.into) == [(0, 10), (10, 10), (20, 10), (30, 10), (40, 7)] and stacks.uploads == []
    assert any(n != threading.current_thread().name for n in threads)                     # dask's worker threads ran the blocks

    # (2) anything else after project -> the generic path under the threaded scheduler: every block decoded + projected ONCE
    other = filled.astype(np.float32) + np.float32(0)
    assert plugin.hip_projection_source(Lazy(other)) is None and plugin.hip_projection_source(Lazy(filled[3:])) is None
    results = {}
    for depth in (0, None, 2):
        decoded.clear(); plan.blocks.clear(); plan.into.clear(); stacks.uploads.clear()
        results[depth] = F.get_piv(Lazy(other), 32, prefetch=depth, **kw)
        assert executor.LAST_STATS["plan"]["source"] == "frames"
        assert decoded == {k: 1 for k in range(5)} and sorted(plan.blocks) == [7, 10, 10, 10, 10] and plan.into == [], (depth, decoded, plan.blocks)
        assert sorted(stacks.uploads) == [(0, 10), (10, 10), (20, 10), (30, 10), (40, 7)]
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(results[0][k], results[None][k], equal_nan=True) and np.array_equal(results[0][k], results[2][k], equal_nan=True), k
        assert np.array_equal(results[0][k], got[k], equal_nan=True), k                      # the hand-off computes the same bits
    assert np.array_equal(results[0].coords["time"], t[1:])
    # the materialised stack: the reference's independent windows
    whole = F.get_piv(OraclePlan(src, dst, *maps)._go(cam), 32, **kw)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(whole[k], got[k], equal_nan=True), k

    # (3) a PIV chunk size that does not divide dask's blocks (user chunksize 17): loads are cut on block boundaries all the same
    decoded.clear(); plan.blocks.clear()
    r17 = F.get_piv(Lazy(other), 32, chunksize=17, **kw)
    assert decoded == {k: 1 for k in range(5)} and sorted(plan.blocks) == [7, 10, 10, 10, 10], (decoded, plan.blocks)
    for k in ("v_x", "v_y", "corr", "s2n"):
        assert np.array_equal(r17[k], got[k], equal_nan=True), k

    # (4) a block that fails inside dask's scheduler: the exception surfaces at ITS chunk, after the chunks before it were launched
    launched = []

    def counting_pairs(fr, ws, ov, thr=None, pair_offset=0, out=None, scale=None):
        launched.append(pair_offset)
        return doubles.oracle_piv_pairs(fr, ws, ov, thr, pair_offset, out, scale)

    mp.setattr(V.piv, "piv_pairs", counting_pairs)

    def broken(block, block_info=None):
        if block_info is not None and block.size and block_info[0]["chunk-location"][0] == 3:
            raise OSError("frame 30 cannot be decoded")
        return block

    bad = da.from_array(np.zeros((47,) + dst, np.float32), chunks=(10,) + dst).map_blocks(broken, dtype=np.float32)
    try:
        F.get_piv(Lazy(bad), 32, **kw)
        raise AssertionError("the loader's exception was swallowed")
    except OSError as exc:
        assert "frame 30" in str(exc)
    assert launched == [0, 10], launched          # pairs [0, 10) and [10, 20) ran (frames 0 .. 29 arrived); nothing beyond the failing block
    mp.undo()
    print("OK dask", dask.__version__)


if __name__ == "__main__":
    main()
