"""The C2 workload with float32 frames (what the orthoprojection with group means hands to get_piv): warm launches, then N timed
ones; one line with ms per launch.  For the profile passes: PROFILE_CMD="python tools/f32_launch.py" bash tools/profile.sh r03_f32

    python tools/f32_launch.py [pairs] [launches]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
H, W = 1080, 1920
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = P + 1
d8, d32, d_o = C.c_void_p(), C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d8), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d32), T * H * W * 4))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * (H // 16) * (W // 16)))
_lib.check(lib.lspiv_synth_particles_dev(d8, T, H, W, 20260927 + 2, 0.02))      # bench.py's stack
step = 64
for t0 in range(0, T, step):
    n = min(step, T - t0)
    a = np.empty((n, H, W), np.uint8)
    _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(a), C.c_void_p(d8.value + t0 * H * W), a.nbytes))
    f = a.astype(np.float32)
    _lib.check(lib.lspiv_memcpy_h2d(C.c_void_p(d32.value + t0 * H * W * 4), _lib.ptr(f), f.nbytes))
lib.lspiv_dev_free(d8)
go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d32, 1, T, H, W, 32, 32, 16, 16, -1.0, d_o, None, None))
for _ in range(6): go()
_lib.check(lib.lspiv_synchronize())
t0 = time.perf_counter()
for _ in range(N): go()
_lib.check(lib.lspiv_synchronize())
ms = (time.perf_counter() - t0) / N * 1e3
print(f'{{"workload": "1080p float32, 32x32 @ 50 %, {P} pairs", "ms_per_step": {ms:.4f}, "pairs_per_s": {P / ms * 1e3:.1f}}}')
