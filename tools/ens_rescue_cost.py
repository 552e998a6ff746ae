"""What the float64 rescue of the ensemble's final fit costs: 1000 pairs of the C2 / C3 shapes accumulated from an HBM-resident stack
(borrowed by the handle), then lspiv_ensemble_finish timed with the rescue on and off (steady state: the second and third call).
usage: ens_rescue_cost.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import DeviceFrames, _lib, piv
lib = _lib.load(); _lib.require_device()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
H, W, T = 1080, 1920, P + 1
d = DeviceFrames.empty((T, H, W), np.uint8)
_lib.check(lib.lspiv_synth_particles_dev(d.c_ptr, T, H, W, 20260927 + 2, 0.02)); _lib.check(lib.lspiv_synchronize())
for ws, ov in ((32, 16), (64, 48)):
    for rescue in (1, 0):
        _lib.set_option("rescue", rescue)
        ens = piv.Ensemble((H, W), (ws, ws), (ov, ov))
        t0 = time.perf_counter(); ens.accumulate(d, 0.2, 3.0); _lib.check(lib.lspiv_synchronize()); acc = time.perf_counter() - t0
        t0 = time.perf_counter(); ens.accumulate(d, 0.2, 3.0); _lib.check(lib.lspiv_synchronize()); acc = time.perf_counter() - t0   # (second call: workspaces exist)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); u, v, cnt = ens.finish(0.2, 1); ts.append((time.perf_counter() - t0) * 1e3)
        st = ens.stats(); ens.close()
        print(f"ensemble {ws}x{ws}, {2 * P} pairs, rescue {rescue}: accumulate {acc * 1e3:.1f} ms / {P} pairs (incl. the corr/s2n block to the host), finish {ts[0]:.2f} {ts[1]:.2f} {ts[2]:.2f} ms, "
              f"{st}, finite {np.isfinite(u).mean():.4f}", flush=True)
_lib.set_option("rescue", 1)
