import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
import pyorc_amd
from oracle import c_oracle
lib = _lib.load()
H, W, P = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 40
T = P + 1
ws, ov = (32, 32), (16, 16)
nr, nc = window.get_array_shape((H, W), ws, ov)
d_frames, d_out = C.c_void_p(), C.c_void_p()
_lib.check(lib.lspiv_dev_malloc(C.byref(d_frames), T * H * W))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_out), 4 * P * nr * nc * 4))
_lib.check(lib.lspiv_synth_particles_dev(d_frames, T, H, W, 123, 0.02))
_lib.check(lib.lspiv_piv_pairs_dev(d_frames, 0, T, H, W, 32, 32, 16, 16, -1.0, d_out, None, None))
_lib.check(lib.lspiv_synchronize())
fr = np.empty((T, H, W), np.uint8)
_lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(fr), d_frames, fr.nbytes))
g = np.empty((4, P, nr, nc), np.float32)
_lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(g), d_out, g.nbytes))
print("frame stats", fr.mean(), fr.max(), (fr == 255).mean())
u, v, cm, sn, cond = c_oracle.piv_pairs(fr, ws, ov, return_cond=True)
ok = c_oracle.well_posed(cond)
print('well-posed fraction', ok.mean())
# host API on the same frames
uh, vh, cmh, snh = pyorc_amd.piv_pairs(fr, ws, ov)
for name, a, b in (("dev-vs-oracle u", g[0], u), ("dev-vs-oracle v", g[1], v), ("dev-vs-host u", g[0], uh), ("dev-vs-oracle cm", g[2], cm), ("dev-vs-oracle sn", g[3], sn)):
    nm = np.isnan(a) != np.isnan(b)
    with np.errstate(all="ignore"):
        e = np.abs(a - b) / np.maximum(np.abs(b), 0.05)
    print(name, "nan mismatch", nm.sum(), "max rel", np.nanmax(e), "count>1e-4", (e > 1e-4).sum(), "| well-posed: nanmm", nm[ok].sum(), "max rel", np.nanmax(e[ok]))
    if nm.sum() or (e > 1e-4).sum():
        bad = np.argwhere(nm | (e > 1e-4))
        print("   first bad", bad[:8].tolist(), "pairs with bad:", np.unique(bad[:, 0])[:20])
print("nan frac oracle", np.isnan(u).mean(), "median u", np.nanmedian(u), "median cm", np.median(cm))
# dump suspicious windows for offline analysis
os.makedirs("gpurun_out", exist_ok=True)
with np.errstate(all="ignore"):
    e_cm = np.abs(g[2] - cm) / np.maximum(np.abs(cm), 0.05)
    e_u = np.abs(g[0] - u) / np.maximum(np.abs(u), 0.05)
bad = np.argwhere((e_cm > 1e-4) | (e_u > 1e-4) | (np.isnan(g[0]) != np.isnan(u)))[:40]
tiles = []
for p, r, c in bad:
    a = fr[p, r*16:r*16+32, c*16:c*16+32]; b = fr[p+1, r*16:r*16+32, c*16:c*16+32]
    tiles.append(np.stack([a, b]))
np.savez_compressed("gpurun_out/bad_tiles.npz", tiles=np.array(tiles), idx=bad, gpu=np.array([g[:, p, r, c] for p, r, c in bad]),
                    ora=np.array([[u[p, r, c], v[p, r, c], cm[p, r, c], sn[p, r, c]] for p, r, c in bad]))
