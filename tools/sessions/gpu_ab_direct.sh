#!/bin/bash
# direct spatial kernel vs LDS-resident DFT passes for the windows neither FFT family serves (non-square, odd 33..63)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/nsq.py <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
H, W, P = 785, 875, 40
T = P + 1
d_f = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W)); _lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 5, 0.03))
for wy, wx in ((24, 16), (32, 24), (40, 24), (48, 32), (64, 32), (33, 33), (35, 35), (41, 41), (49, 49), (63, 63), (17, 17), (19, 19)):
    oy, ox = wy // 2, wx // 2
    nr, nc = window.get_array_shape((H, W), (wy, wx), (oy, ox))
    d_o = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
    go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d_f, 0, T, H, W, wy, wx, oy, ox, -1.0, d_o, None, None))
    go(); _lib.check(lib.lspiv_synchronize())
    t0 = time.perf_counter()
    for _ in range(3): go()
    _lib.check(lib.lspiv_synchronize()); t = (time.perf_counter() - t0) / 3
    print(f"kind {lib.lspiv_kernel_kind(wy, wx)} window {wy}x{wx}: {P/t:9.0f} pairs/s {P*nr*nc/t/1e6:8.2f} Mwin/s", flush=True)
    lib.lspiv_dev_free(d_o)
PY
echo "== direct"; LSPIV_DFT_MIN_AREA=1000000 python /tmp/nsq.py; echo "== dft passes"; LSPIV_DFT_MIN_AREA=1 python /tmp/nsq.py
