"""Throughput per window size on the Ngwerere geometry (785 x 875): every even size 6..64 (pyorc's recipes: 25 -> 24,
tests: 10..20) and two odd ones; LSPIV_NO_PFA=1 / LSPIV_NO_EMBED=1 select the embedded / direct kernels instead."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
H, W, P = 785, 875, int(sys.argv[1]) if len(sys.argv) > 1 else 40
T = P + 1
d_f = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 5, 0.02))
for ws in (6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32, 34, 35, 36, 40, 42, 44, 48, 50, 52, 56, 60, 62, 64):
    ov = ws // 2
    nr, nc = window.get_array_shape((H, W), (ws, ws), (ov, ov))
    d_o = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
    go = lambda: _lib.check(lib.lspiv_piv_pairs_dev(d_f, 0, T, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
    go(); _lib.check(lib.lspiv_synchronize())
    t0 = time.perf_counter()
    for _ in range(3): go()
    _lib.check(lib.lspiv_synchronize()); dt = (time.perf_counter() - t0) / 3
    kind = {1: "fft32", 2: "fft64", 3: "direct", 4: "embed32", 5: "embed64", 6: "fft8/16", 7: "embed16", 8: "pfa-fft"}[lib.lspiv_kernel_kind(ws, ws)]
    macs = P * nr * nc * float(ws) ** 4
    print(f"785x875 win {ws}/{ov} ({kind}): {nr*nc} windows/pair, {dt*1e3:.2f} ms / {P} pairs -> {P/dt:.0f} pairs/s, {P*nr*nc/dt/1e6:.1f} Mvec/s"
          + (f", {macs/dt/1e12:.2f} TMAC/s" if kind == "direct" else ""), flush=True)
    lib.lspiv_dev_free(d_o)
