"""Bits of an ensemble accumulation on the 1080p synthetic stack: sha256 of corr_sum / corr_count / the masked per-pair (corr_max, s2n) /
u / v, for comparing two builds of the library (LSPIV_LIBRARY=<other build>) that must agree bit for bit.

    python tools/ens_hash.py [window] [overlap] [pairs] [signal_threshold|-1]"""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, piv
lib = _lib.load(); _lib.require_device()
H, W = 1080, 1920
ws = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 48
P = int(sys.argv[3]) if len(sys.argv) > 3 else 120
thr = float(sys.argv[4]) if len(sys.argv) > 4 else -1.0
T = P + 1
ens = piv.Ensemble((H, W), (ws, ws), (ov, ov))
n_win = ens.n_rows * ens.n_cols
d_f, d_o = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_f), T * H * W)); _lib.check(lib.lspiv_dev_malloc(C.byref(d_o), 8 * P * n_win))
_lib.check(lib.lspiv_synth_particles_dev(d_f, T, H, W, 20260927 + 2, 0.02))
ens.set_retain(ens.RETAIN_BORROW)
ens.accumulate_dev(d_f.value, np.uint8, T, 0.2, 3.0, d_o.value, None if thr < 0 else thr)
_lib.check(lib.lspiv_synchronize())
cs = np.empty((2, P, n_win), np.float32)
_lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(cs), d_o, cs.nbytes))
s, k = ens.export_state()
u, v, cnt = ens.finish(0.2, 1)
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
if os.environ.get("ENS_DUMP"):   # the arrays themselves, for tools/ens_diff.py
    np.savez(os.environ["ENS_DUMP"], corr_sum=s, count=k, cmax_s2n=cs, u=u, v=v)
print(f"ens_hash {ws}/{ov} P={P} thr={thr}: corr_sum {h(s)} count {h(k)} cmax_s2n {h(cs)} u {h(u)} v {h(v)} kept {float(k.mean()):.2f} finite {float(np.isfinite(u).mean()):.4f}")
