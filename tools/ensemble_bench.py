"""Ensemble-correlation mode (A10, pyorc/velocimetry/ffpiv.py:182-376) on a device-resident 1080p uint8 stack: every pair's
correlation plane is added to the running (n_win, wy, wx) sum in HBM.  Pairs/s and the read-modify-write traffic of the sum."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, piv
lib = _lib.load(); _lib.require_device()
H, W, T = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 501
for ws, ov in (((32, 32), (16, 16)), ((64, 64), (48, 48)), ((24, 24), (12, 12)), ((16, 16), (8, 8))):
    ens = piv.Ensemble((H, W), ws, ov)
    n_win = ens.n_rows * ens.n_cols
    d_f, d_o = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 8 * (T - 1) * n_win))
    _lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 7, 0.02))
    run = lambda: ens.accumulate_dev(d_f.value, np.uint8, T, 0.2, 3.0, d_o.value)
    run(); _lib.check(lib.lspiv_synchronize())
    t0 = time.perf_counter()
    for _ in range(3): run()
    _lib.check(lib.lspiv_synchronize()); t = (time.perf_counter() - t0) / 3
    u, v, cnt = ens.finish(0.2, 4 * (T - 1))
    rmw = 2 * 4 * n_win * ws[0] * ws[1] * ((T - 1) // 2)   # 64 x 64: one read + one write of the partial sum per TWO pairs
    note = f"corr_sum read-modify-write {rmw/t/1e9:.0f} GB/s" if ws[0] > 32 else "partial sums accumulated in registers"
    print(f"ensemble {ws[0]}x{ws[1]}: {T-1} pairs in {t*1e3:.1f} ms = {(T-1)/t:.0f} pairs/s; {note} "
          f"(sum is {4*n_win*ws[0]*ws[1]/1e6:.0f} MB); finite vectors {np.isfinite(u).mean():.3f}, median u {np.nanmedian(u):.2f}")
    ens.close(); lib.lspiv_dev_free(d_f); lib.lspiv_dev_free(d_o)
