"""One rank of the GPU sharding tests (tests/test_gpu_shard.py starts `world` of these on device 0; they talk over the
shared-memory transport of the C ABI -- RCCL refuses two ranks on one GPU).  usage: shard_gpu_worker.py <mode> <out_dir> <n_frames>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyorc_amd import DeviceFrames, _lib, piv, shard, window  # noqa: E402
from pyorc_amd.comm import Comm  # noqa: E402
from pyorc_amd.synth import particle_stack  # noqa: E402

mode, out_dir, n_frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
WS, OV = (32, 32), (16, 16)
_lib.require_device()
_lib.check(_lib.load().lspiv_set_device(0))
comm = Comm(rank, world, transport="shm", id_file=os.path.join(out_dir, "comm_id"), timeout=120)
try:
    stack = particle_stack(n_frames, 96, 128, seed=77, density=0.03)
    n_pairs = n_frames - 1
    touched = []

    def load(a, b):
        touched.append((a, b))
        return stack[a:b]

    if mode == "piv":           # host arrays in, the library's default compute (piv.piv_pairs), default alignment (25)
        full = shard.sharded_piv(load, n_pairs, WS, OV, comm, frame_shape=stack.shape[1:])   # (the grid's anchor length: 25 on these small frames)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full, touched=np.array(touched), align=window.chunk_alignment(WS, stack.shape[1:], OV))
    elif mode == "piv_dev":     # the rank's block resident in HBM, results gathered between devices
        plan = shard.ShardedPivDev(comm, n_pairs, stack.shape[1:], WS, OV, record_timings=True)
        f0, f1 = plan.frame_block()
        block = DeviceFrames.from_host(stack[f0:f1]) if f1 > f0 else DeviceFrames.empty((0,) + stack.shape[1:], np.uint8)
        for _ in range(3):      # three pipelined steps: buffers are reused, the last gather must still be the full result
            plan.step(block)
        full = plan.gathered_host()
        tm = plan.timings()
        plan.close()
        one = shard.sharded_piv_dev(block, n_pairs, WS, OV, comm)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), full=full, one=one, block=np.array([f0, f1]), kernel_ms=np.array(tm["kernel_ms"]),
                 gather_ms=np.array(tm["gather_ms"]))
    elif mode == "ensemble":
        u, v, cnt, cm, sn = shard.sharded_ensemble(load, n_pairs, lambda: piv.Ensemble(stack.shape[1:], WS, OV), 0.1, 1.5, 0.2, comm,
                                                   n_chunks=1, align=window.chunk_alignment(WS, stack.shape[1:], OV))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), u=u, v=v, cnt=cnt, cm=cm, sn=sn, touched=np.array(touched))
    else:
        raise SystemExit(f"unknown mode {mode}")
finally:
    comm.close()
