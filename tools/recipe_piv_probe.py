"""What the PIV launch of the reference's recipe costs on its own data: edge-detected, clipped float32 frames on the ortho grid, with
the rescue pass on and off, and the rescue counters.   usage: recipe_piv_probe.py [pairs]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
from pyorc_amd.project import Projection
from pyorc_amd.synth import projection_maps
lib = _lib.load(); _lib.require_device()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T, H, W, HO, WO = P + 1, 1080, 1920, 810, 1440
def alloc(n):
    p = C.c_void_p(); _lib.check(lib.lspiv_dev_malloc(C.byref(p), n)); return p
d_cam, d_n, d_e, d_o = alloc(T * H * W), alloc(T * H * W), alloc(T * H * W * 4), alloc(T * HO * WO * 4)
_lib.check(lib.lspiv_synth_particles_dev(d_cam, T, H, W, 3, 0.02))
_lib.check(lib.lspiv_normalize_dev(d_cam, T, H, W, 15, d_n, None))
_lib.check(lib.lspiv_edge_detect_clip_dev(d_n, 0, T, H, W, 3, 5, -5.0, 5.0, d_e, None))
p = Projection((H, W), (HO, WO), *projection_maps((H, W), (HO, WO), tilt=0.1, seed=1))
p.project_frames_dev(d_e.value, np.float32, T, d_o.value)
nr, nc = window.get_array_shape((HO, WO), (32, 32), (16, 16))
d_r = alloc(16 * P * nr * nc)
def go(src, dt): _lib.check(lib.lspiv_piv_pairs_dev(src, dt, T, HO, WO, 32, 32, 16, 16, -1.0, d_r, None, None))
def timed(src, dt):
    for _ in range(3): go(src, dt)
    _lib.check(lib.lspiv_synchronize()); t0 = time.perf_counter()
    for _ in range(10): go(src, dt)
    _lib.check(lib.lspiv_synchronize()); return (time.perf_counter() - t0) / 10 * 1e3
st = (C.c_int64 * 8)()
for rescue in ((int(os.environ.get("LSPIV_RESCUE", 1)),) if os.environ.get("PROBE_ONE") else (1, 0, 1, 0)):   # (alternating: the first timing of a process runs on ramping clocks)
    _lib.check(lib.lspiv_set_option(b"rescue", rescue))
    ms = timed(d_o, 1)
    _lib.check(lib.lspiv_rescue_stats(None, st))
    print(f"recipe frames (edge-detected, clipped, projected float32) {HO}x{WO} P={P} rescue={rescue}: {ms:.3f} ms  stats {list(st)[:5]} of {P*nr*nc} windows")
_lib.check(lib.lspiv_set_option(b"rescue", 1))
