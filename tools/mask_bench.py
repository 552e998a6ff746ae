"""N3 measurement: the Ngwerere mask recipe on a device-resident C2 result block (1000 pairs x 66 x 119 vectors):
scale to m/s -> corr, minmax, rolling, outliers, variance, count masks (each applied in place) -> int16 pack.
Compares with the numpy oracle of the same chain on the host."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
T, R, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 66, 119
n, N = R * Cc, T * R * Cc
rng = np.random.default_rng(0)
f = np.empty((4, T, R, Cc), np.float32)
f[0] = rng.normal(3.0, 1.5, (T, R, Cc)); f[1] = rng.normal(0.0, 1.0, (T, R, Cc)); f[2] = rng.random((T, R, Cc)); f[3] = rng.random((T, R, Cc)) * 30
f[:, rng.random((T, R, Cc)) < 0.1] = np.nan
dt = np.full(T, 1 / 30.0)
d_f, d_w, d_m, d_p = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
for d, b in ((d_f, 16 * N), (d_w, 16 * N), (d_m, N), (d_p, 8 * N)):
    _lib.check(lib.lspiv_dev_malloc(C.byref(d), b))
_lib.check(lib.lspiv_memcpy_h2d(d_f, _lib.ptr(f), f.nbytes))
recipe = ((3, [0.1], 1), (0, [0.1, 5.0], 1), (7, [5, 0.5], 1), (5, [1.0, 0], 1), (6, [5, 1], 0), (2, [0.33], 0))
names = {0: "minmax", 2: "count", 3: "corr", 5: "outliers", 6: "variance", 7: "rolling", 8: "window_nan", 9: "window_mean"}
def sync(): _lib.check(lib.lspiv_synchronize())
def chain():
    _lib.check(lib.lspiv_memcpy_h2d(d_w, _lib.ptr(f), 0))  # no-op copy keeps the call pattern; the block is restored below
    lib.lspiv_scale_velocity_dev(d_w, T, n, 0.01, -0.01, _lib.ptr(dt), None)
    for kind, params, has_t in recipe:
        p = np.asarray(params, np.float64)
        _lib.check(lib.lspiv_mask_dev(d_w, T, R, Cc, kind, _lib.ptr(p), len(p), d_m, None))
        _lib.check(lib.lspiv_mask_apply_dev(d_w, T, R, Cc, d_m, has_t, None))
    _lib.check(lib.lspiv_pack_int16_dev(d_w, 4 * N, 0.01, -9999, d_p, None))
# per-kernel timings
import ctypes
hipcpy = lambda: _lib.check(lib.lspiv_memcpy_h2d(d_w, _lib.ptr(f), f.nbytes))
hipcpy(); sync()
for kind, params, has_t in recipe + ((8, [0.7, -1, 1, -1, 1], 1), (9, [0.7, 0, -1, 1, -1, 1], 1)):
    p = np.asarray(params, np.float64)
    run = lambda: _lib.check(lib.lspiv_mask_dev(d_w, T, R, Cc, kind, _lib.ptr(p), len(p), d_m, None))
    run(); sync(); t0 = time.perf_counter()
    for _ in range(10): run()
    sync(); t = (time.perf_counter() - t0) / 10
    rd = {0: 8, 2: 4, 3: 4, 5: 16, 6: 16, 7: 8, 8: 4, 9: 8}[kind] * N + (N if has_t else n)
    print(f"mask {names[kind]:12s}: {t*1e3:6.3f} ms, {rd/t/1e9:6.0f} GB/s algorithmic")
hipcpy(); sync(); t0 = time.perf_counter(); chain(); sync(); t_dev = time.perf_counter() - t0
print(f"device chain (scale + 6 masks applied + int16 pack) on {T} x {R} x {Cc}: {t_dev*1e3:.2f} ms = {T/t_dev:.0f} time steps/s")
if T <= 1000:
    from oracle import mask_oracle as mo, piv_oracle as po
    t0 = time.perf_counter()
    g = f.copy(); g[0], g[1] = mo.scale_velocity(f[0], f[1], 0.01, -0.01, dt)
    for fn in (mo.corr, mo.minmax, mo.rolling, mo.outliers, mo.variance, mo.count):
        g = mo.apply(g, fn(g))
    ref = po.encode_int16(g); t_cpu = time.perf_counter() - t0
    pk = np.empty((4, T, R, Cc), np.int16); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(pk), d_p, pk.nbytes))
    print(f"numpy oracle chain: {t_cpu*1e3:.0f} ms ({t_cpu/t_dev:.0f}x), packed results identical: {np.array_equal(pk, ref)}")
