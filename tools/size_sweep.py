"""How the device-resident PIV launch scales DOWN: pairs/s and window-pairs/s over the number of pairs and the frame size (what a
typical pyorc run hands over is a few hundred frames of ~1000 x 1000, not BASELINE's 1000 pairs of 1080p).  Timed by wall clock around
`reps` back-to-back launches on a resident stack, rescue pass on.    usage: size_sweep.py [dtype 0|1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
dtype = int(sys.argv[1]) if len(sys.argv) > 1 else 0
es = (1, 4)[dtype]

def run(H, W, P, ws, ov):
    T = P + 1
    nr, nc = window.get_array_shape((H, W), (ws, ws), (ov, ov))
    d_u8, d_f, d_o = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_u8), T * H * W))
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
    _lib.check(lib.lspiv_synth_particles_dev(d_u8, T, H, W, 5, 0.02))
    d_in = d_u8
    if dtype == 1:                                   # float32 frames: the edge-detected stack
        _lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W * 4))
        _lib.check(lib.lspiv_edge_detect_dev(d_u8, 0, T, H, W, 3, 5, d_f, None))
        d_in = d_f
    def go(): _lib.check(lib.lspiv_piv_pairs_dev(d_in, dtype, T, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
    for _ in range(3): go()
    _lib.check(lib.lspiv_synchronize())
    reps = max(3, min(200, int(2000 / P)))
    t0 = time.perf_counter()
    for _ in range(reps): go()
    _lib.check(lib.lspiv_synchronize())
    dt = (time.perf_counter() - t0) / reps
    print(f"{H}x{W} {('u8','f32')[dtype]} win {ws}/{ov} P={P:5d}: {dt*1e3:8.3f} ms/launch  {P/dt:9.0f} pairs/s  {P*nr*nc/dt/1e6:8.1f} M window-pairs/s", flush=True)
    for p in (d_u8, d_f, d_o):
        if p: lib.lspiv_dev_free(p)

for H, W in ((1080, 1920), (810, 1440), (785, 875), (540, 960)):
    for P in (20, 50, 100, 200, 400, 1000):
        run(H, W, P, 32, 16)
run(810, 1440, 200, 64, 48); run(810, 1440, 50, 64, 48)
