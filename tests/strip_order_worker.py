"""One process of tests/test_gpu_strip_order.py: PIV of a seeded stack under whatever LSPIV_STRIP_W the environment holds (the library
reads it once per process).  usage: strip_order_worker.py <out.npz> <window> <overlap> <dtype> <H> <W> <pairs> <0 per time step | 1 ensemble | 2 per time step with planes> [signal threshold]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyorc_amd import _lib, piv  # noqa: E402
from pyorc_amd.synth import particle_stack  # noqa: E402

thr = float(sys.argv[9]) if len(sys.argv) > 9 else None   # signal threshold: the kernels' WANT_NZ variants
out, ws, ov, dt, H, W, P, ens = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), np.dtype(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
_lib.require_device()
stack = particle_stack(P + 1, H, W, seed=4242, density=0.03)
if dt != np.uint8:
    stack = stack.astype(dt) * 0.7 - 11.0
if ens:
    e = piv.Ensemble((H, W), (ws, ws), (ov, ov))
    cm, sn = e.accumulate(stack, 0.1, 1.5, thr)
    u, v, cnt = e.finish(0.2, 1)
    e.close()
    np.savez(out, u=u, v=v, cnt=cnt, cm=cm, sn=sn)
elif ens == 2:   # the correlation planes too (cross_corr's volume): their digest and per-window sums
    import hashlib

    u, v, cm, sn, planes = piv.piv_pairs(stack, (ws, ws), (ov, ov), thr, return_planes=True)
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(planes).tobytes()).digest(), dtype=np.uint32).copy()
    np.savez(out, u=u, v=v, cm=cm, sn=sn, digest=digest, sums=planes.reshape(planes.shape[0], planes.shape[1], -1).sum(axis=-1))
else:
    u, v, cm, sn = piv.piv_pairs(stack, (ws, ws), (ov, ov), thr)
    np.savez(out, u=u, v=v, cm=cm, sn=sn)
