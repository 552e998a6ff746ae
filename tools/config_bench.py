"""Throughput of the other BASELINE.json configurations + ensemble mode (device-resident stacks)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window, piv
lib = _lib.load(); _lib.require_device()

def run(H, W, P, ws, ov, reps=5, seed=5):
    T = P + 1
    nr, nc = window.get_array_shape((H, W), (ws, ws), (ov, ov))
    d_f, d_o = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W))
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
    _lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, seed, 0.02))
    def go(): _lib.check(lib.lspiv_piv_pairs_dev(d_f, 0, T, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
    go(); _lib.check(lib.lspiv_synchronize())
    t0 = time.perf_counter()
    for _ in range(reps): go()
    _lib.check(lib.lspiv_synchronize())
    dt = (time.perf_counter() - t0) / reps
    b_alg = 2 * H * W + 16 * nr * nc
    print(f"{H}x{W} win {ws}/{ov} P={P}: {dt*1e3:.2f} ms -> {P/dt:.0f} pairs/s, {P/dt*nr*nc/1e6:.1f} Mvec/s, "
          f"B_alg {b_alg/1e6:.3f} MB/pair -> {P/dt*b_alg/1e9:.1f} GB/s ({P/dt*b_alg/8e12*100:.2f}% of 8 TB/s)", flush=True)
    fr = None
    if P <= 200:
        fr = np.empty((T, H, W), np.uint8); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(fr), d_f, fr.nbytes))
    lib.lspiv_dev_free(d_f); lib.lspiv_dev_free(d_o)
    return fr

run(1080, 1920, 1000, 32, 16)
run(1080, 1920, 1000, 64, 48, reps=3)
run(2160, 3840, 1000, 32, 16, reps=3)
run(785, 875, 20, 32, 16)
fr = run(1080, 1920, 200, 32, 16)
# ensemble mode, host-fed (the API keeps corr_sum in HBM across chunks)
for ws, ov in ((32, 16), (64, 48)):
    ens = piv.Ensemble(fr.shape[1:], (ws, ws), (ov, ov))
    ens.accumulate(fr[:3], 0.2, 3.0)
    t0 = time.perf_counter(); ens.accumulate(fr, 0.2, 3.0); dt = time.perf_counter() - t0
    u, v, cnt = ens.finish(0.2, 1)
    print(f"ensemble {ws}/{ov}: 200 pairs in {dt*1e3:.1f} ms -> {200/dt:.0f} pairs/s (host-fed), valid {np.isfinite(u).mean():.2f}")
    ens.close()
