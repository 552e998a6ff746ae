import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyorc_amd import _lib
lib = _lib.load(); _lib.require_device()
H, W, T = 1080, 1920, 101
n = H * W
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * n)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), T * n * 4))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 3, 0.02))
def timed(fn, reps=5):
    for _ in range(2): fn()
    _lib.check(lib.lspiv_synchronize()); t0 = time.perf_counter()
    for _ in range(reps): fn()
    _lib.check(lib.lspiv_synchronize()); return (time.perf_counter() - t0) / reps
out = []
for k in (23, 27, 31): out.append("k=%d %.2f" % (k, 1e3 * timed(lambda: _lib.check(lib.lspiv_gaussian_blur_dev(d_f, 0, T, H, W, k, d_o, None)))))
for a, b in ((23, 31), (13, 25), (9, 23)): out.append("edge %d|%d %.2f" % (a, b, 1e3 * timed(lambda: _lib.check(lib.lspiv_edge_detect_dev(d_f, 0, T, H, W, a, b, d_o, None)))))
print(os.environ.get("TAG"), "ms per %d frames:" % T, " | ".join(out))
