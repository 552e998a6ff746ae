"""Bits of a per-timestep launch on the synthetic 1080p stack (sha256 of the [u | v | corr | s2n] block): for comparing job orders / builds.
    python tools/piv_hash.py [window] [overlap] [pairs] [H] [W]"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
lib = _lib.load(); _lib.require_device()
ws = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 16
P = int(sys.argv[3]) if len(sys.argv) > 3 else 130
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
W = int(sys.argv[5]) if len(sys.argv) > 5 else 1920
nr, nc = window.get_array_shape((H, W), (ws, ws), (ov, ov))
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), (P + 1) * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 16 * P * nr * nc))
_lib.check(lib.lspiv_synth_particles_dev(d_f, P + 1, H, W, 20260927 + 2, 0.02))
_lib.check(lib.lspiv_piv_pairs_dev(d_f, 0, P + 1, H, W, ws, ws, ov, ov, -1.0, d_o, None, None))
out = np.empty((4, P, nr, nc), np.float32)
_lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(out), d_o, out.nbytes))
print(f"piv_hash {ws}/{ov} P={P} {H}x{W}: {hashlib.sha256(out.tobytes()).hexdigest()[:16]} finite {float(np.isfinite(out[0]).mean()):.4f}")
