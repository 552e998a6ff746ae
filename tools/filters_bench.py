"""N2 measurement: element-wise filters on a device-resident 1080p uint8 stack (HBM-bound streaming kernels)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
H, W, T = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 401
n = H * W
d_f, d_o, d_o2, d_n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * n)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), (T - 1) * n * 4))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_o2), (T - 1) * n * 4)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_n), T * n))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 3, 0.02))
def timed(fn, reps=5):
    fn(); _lib.check(lib.lspiv_synchronize()); t0 = time.perf_counter()
    for _ in range(reps): fn()
    _lib.check(lib.lspiv_synchronize()); return (time.perf_counter() - t0) / reps
t = timed(lambda: _lib.check(lib.lspiv_time_diff_dev(d_f, 0, T, H, W, 0.0, 0, d_o, None)))
b = (T - 1) * n * (2 * 1 + 4)
print(f"time_diff  u8->f32: {t*1e3:.2f} ms / {T-1} frames, {b/t/1e9:.0f} GB/s algorithmic (2 u8 reads + 1 f32 write per px) = {b/t/8e12*100:.1f}% of 8 TB/s")
t = timed(lambda: _lib.check(lib.lspiv_minmax_dev(d_o, (T - 1) * n, -5.0, 5.0, d_o2, None)))
b = (T - 1) * n * 8
print(f"minmax     f32    : {t*1e3:.2f} ms, {b/t/1e9:.0f} GB/s = {b/t/8e12*100:.1f}% of 8 TB/s")
t = timed(lambda: _lib.check(lib.lspiv_normalize_dev(d_f, T, H, W, 15, d_n, None)), 3)
b = T * n * (1 + 1 + 1)  # min/max pass read, stretch pass read + write (the mean plane sits in registers)
print(f"normalize  u8->u8 : {t*1e3:.2f} ms / {T} frames, {b/t/1e9:.0f} GB/s algorithmic = {b/t/8e12*100:.1f}% of 8 TB/s (3 passes: sampled mean, per-frame min/max, stretch)")
t = timed(lambda: _lib.check(lib.lspiv_time_range_dev(d_f, 0, T, H, W, d_n, None)))
b = T * n + n
print(f"range      u8->u8 : {t*1e3:.2f} ms / {T} frames, {b/t/1e9:.0f} GB/s algorithmic (every frame read once) = {b/t/8e12*100:.1f}% of 8 TB/s")
for name, fn, k in (("smooth k=3", lambda: lib.lspiv_gaussian_blur_dev(d_f, 0, T - 1, H, W, 3, d_o, None), 3),
                    ("smooth k=7", lambda: lib.lspiv_gaussian_blur_dev(d_f, 0, T - 1, H, W, 7, d_o, None), 7),
                    ("edge 3|5   ", lambda: lib.lspiv_edge_detect_dev(d_f, 0, T - 1, H, W, 3, 5, d_o, None), 5),
                    ("edge 5|9 (wdw 2|4)  ", lambda: lib.lspiv_edge_detect_dev(d_f, 0, T - 1, H, W, 5, 9, d_o, None), 9),
                    ("smooth k=11", lambda: lib.lspiv_gaussian_blur_dev(d_f, 0, T - 1, H, W, 11, d_o, None), 11),
                    ("edge 13|21 (wdw 6|10)", lambda: lib.lspiv_edge_detect_dev(d_f, 0, T - 1, H, W, 13, 21, d_o, None), 21),
                    ("edge 5|15  ", lambda: lib.lspiv_edge_detect_dev(d_f, 0, T - 1, H, W, 5, 15, d_o, None), 15)):
    t = timed(lambda: _lib.check(fn()))
    b = (T - 1) * n * (1 + 4)
    print(f"{name} u8->f32: {t*1e3:.2f} ms / {T-1} frames = {(T-1)/t:.0f} frames/s, {b/t/1e9:.0f} GB/s algorithmic (1 u8 read + 1 f32 write per px) = {b/t/8e12*100:.1f}% of 8 TB/s")
for samples in (5, 25):
    t = timed(lambda: _lib.check(lib.lspiv_reduce_rolling_dev(d_f, T, H, W, samples, d_n, None)), 3)
    b = T * n * 2
    print(f"reduce_rolling(samples={samples}) u8->u8: {t*1e3:.2f} ms / {T} frames, {b/t/1e9:.0f} GB/s algorithmic (read once + written once credited; two passes) = {b/t/8e12*100:.1f}% of 8 TB/s")
