"""Device-resident throughput per frame dtype (uint8 / float32 / float64), 1080p: all stacks resident at once, launches
interleaved in rounds so that every dtype meets the same clocks (a stack uploaded between two timings leaves the GPU idle for
seconds, and the first launches after that run visibly slower -- round 3 found the float32 rate understated by 20 % that way).

    python tools/dtype_bench.py [pairs] [rescue 0|1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
H, W, P = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2: _lib.check(lib.lspiv_set_option(b"rescue", int(sys.argv[2])))
T = P + 1
d8, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d8), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * (H // 16) * (W // 16)))
_lib.check(lib.lspiv_synth_particles_dev(d8, T, H, W, 5, 0.02))
fr = np.empty((T, H, W), np.uint8); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(fr), d8, fr.nbytes))
stacks = {"uint8": (0, d8)}
for code, dt in ((1, np.float32),) + (() if os.environ.get("NO_F64") else ((2, np.float64),)):
    d = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d), T * H * W * np.dtype(dt).itemsize))
    step = 64                                   # converted and uploaded in slabs: no second full-size host copy
    for t0 in range(0, T, step):
        a = fr[t0:t0 + step].astype(dt)
        _lib.check(lib.lspiv_memcpy_h2d(C.c_void_p(d.value + t0 * H * W * a.itemsize), _lib.ptr(a), a.nbytes))
    stacks[np.dtype(dt).name] = (code, d)
del fr
for ws, ov in ((32, 16),) + (() if os.environ.get("ONLY_32") else ((64, 48),)):
    best = {k: 1e9 for k in stacks}
    for rnd in range(4):
        for name, (code, d) in stacks.items():
            go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d, code, T, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
            go(); _lib.check(lib.lspiv_synchronize())
            t0 = time.perf_counter()
            for _ in range(3): go()
            _lib.check(lib.lspiv_synchronize())
            best[name] = min(best[name], (time.perf_counter() - t0) / 3)
    for name, dtm in best.items():
        print(f"{name:8s} win {ws}/{ov}: {dtm*1e3:7.2f} ms / {P} pairs -> {P/dtm:8.0f} pairs/s ({best['uint8']/dtm*100:5.1f} % of uint8)", flush=True)
