"""Host-fed call cost against chunk size: lspiv_piv_pairs_at on float64 / uint8 host chunks of 26 ... 201 frames (1080p), best of 5,
plus the same through get_piv with that chunk size -- what a chunked run pays per chunk beyond the PCIe transfer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, piv, frames as F
from pyorc_amd.synth import particle_stack
lib = _lib.load(); _lib.require_device()
base = particle_stack(201, 1080, 1920, seed=1)
for dt in (np.float64, np.uint8):
    arr = base.astype(dt)
    for n in (26, 51, 101, 201):
        a = arr[:n]
        piv.piv_pairs(a, (32, 32), (16, 16))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); piv.piv_pairs(a, (32, 32), (16, 16)); ts.append(time.perf_counter() - t0)
        fresh = []
        for _ in range(3):
            b = np.empty_like(a); b[:] = a          # a freshly allocated chunk, like a materialised dask block
            t0 = time.perf_counter(); piv.piv_pairs(b, (32, 32), (16, 16)); fresh.append(time.perf_counter() - t0)
            del b
        print(f"{np.dtype(dt).name:8s} {n:4d} frames: call {min(ts) * 1e3:7.2f} ms = {(n - 1) / min(ts):8.0f} pairs/s | fresh array {min(fresh) * 1e3:7.2f} ms | PCIe alone at 52 GB/s {a.nbytes / (2 if dt == np.float64 else 1) / 52e9 * 1e3:6.2f} ms")
    t = np.arange(201) / 30.0
    for cs in (26, 101):
        F.get_piv(arr, 32, time=t, resolution=0.01, chunksize=cs)
        t0 = time.perf_counter(); F.get_piv(arr, 32, time=t, resolution=0.01, chunksize=cs); d = time.perf_counter() - t0
        print(f"{np.dtype(dt).name:8s} get_piv chunksize {cs:3d}: {d * 1e3:7.1f} ms for 200 pairs = {200 / d:7.0f} pairs/s")
