"""Device-resident throughput per frame dtype (uint8 / float32 / float64), 1080p, 32x32 @ 50 %."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
H, W, P = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 300
T = P + 1
nr, nc = window.get_array_shape((H, W), (32, 32), (16, 16))
d8, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d8), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
_lib.check(lib.lspiv_synth_particles_dev(d8, T, H, W, 5, 0.02))
fr = np.empty((T, H, W), np.uint8); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(fr), d8, fr.nbytes))
for code, dt in ((0, np.uint8), (1, np.float32), (2, np.float64)):
    for ws, ov in ((32, 16), (64, 48)):
        a = fr.astype(dt)
        d = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d), a.nbytes)); _lib.check(lib.lspiv_memcpy_h2d(d, _lib.ptr(a), a.nbytes))
        go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d, code, T, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
        go(); _lib.check(lib.lspiv_synchronize())
        t0 = time.perf_counter()
        for _ in range(3): go()
        _lib.check(lib.lspiv_synchronize()); dtm = (time.perf_counter() - t0) / 3
        print(f"{np.dtype(dt).name:8s} win {ws}/{ov}: {dtm*1e3:7.2f} ms / {P} pairs -> {P/dtm:8.0f} pairs/s", flush=True)
        lib.lspiv_dev_free(d)
