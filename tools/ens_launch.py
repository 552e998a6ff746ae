"""Ensemble-correlation launches on an HBM-resident 1080p uint8 stack (bench.py's synthetic stack), for the profile passes:
PROFILE_CMD="python tools/ens_launch.py 64 48 1000 6" bash tools/profile.sh r04_ens64

    python tools/ens_launch.py [window] [overlap] [pairs] [launches] [u8|f32] [signal_threshold|-1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, piv
lib = _lib.load(); _lib.require_device()
H, W = 1080, 1920
ws = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ov = int(sys.argv[2]) if len(sys.argv) > 2 else ws // 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dt = sys.argv[5] if len(sys.argv) > 5 else "u8"
thr = float(sys.argv[6]) if len(sys.argv) > 6 else -1.0
T = P + 1
ens = piv.Ensemble((H, W), (ws, ws), (ov, ov))
n_win = ens.n_rows * ens.n_cols
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 8 * P * n_win))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 20260927 + 2, 0.02))
d_in, np_dt = d_f.value, np.uint8
if dt == "f32":      # the same samples as float32 frames (what a projection with group means hands over)
    from pyorc_amd.device import DeviceFrames
    host = np.empty((T, H, W), np.uint8)
    _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(host), d_f, host.nbytes))
    dev32 = DeviceFrames.from_host(host.astype(np.float32))
    d_in, np_dt = dev32.ptr, np.float32
go = lambda: ens.accumulate_dev(d_in, np_dt, T, 0.2, 3.0, d_o.value, None if thr < 0 else thr)
for _ in range(3): go()
_lib.check(lib.lspiv_synchronize())
t0 = time.perf_counter()
for _ in range(N): go()
_lib.check(lib.lspiv_synchronize())
ms = (time.perf_counter() - t0) / N * 1e3
t0 = time.perf_counter()
u, v, cnt = ens.finish(0.2, 1)
fin_ms = (time.perf_counter() - t0) * 1e3
print(f'{{"workload": "ensemble 1080p {dt}, {ws}x{ws} @ overlap {ov}, {P} pairs, threshold {thr}", "ms_per_step": {ms:.4f}, "pairs_per_s": {P / ms * 1e3:.1f}, '
      f'"finish_ms": {fin_ms:.2f}, "finite": {float(np.isfinite(u).mean()):.4f}}}')
