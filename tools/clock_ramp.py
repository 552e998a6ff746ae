"""How long after an idle period do launches run slower?  Per-launch wall times of the C2 workload after 3 s of idle."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
H, W, P = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = P + 1
d8, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d8), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * (H // 16) * (W // 16)))
_lib.check(lib.lspiv_synth_particles_dev(d8, T, H, W, 5, 0.02))
go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d8, 0, T, H, W, 32, 32, 16, 16, -1.0, d_o, None, None))
go(); _lib.check(lib.lspiv_synchronize())
for idle in (3.0, 0.5, 0.05):
    time.sleep(idle)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); go(); _lib.check(lib.lspiv_synchronize()); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"after {idle} s idle:", " ".join(f"{t:.2f}" for t in ts), flush=True)
