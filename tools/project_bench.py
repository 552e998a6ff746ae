"""N1 measurement: orthoprojection gather kernel (camera 1080p uint8 -> ortho float32) and the device-resident chain
camera frames -> project -> PIV.  Reports frames/s, algorithmic GB/s (source frame read once + ortho frame written once)
against the 8 TB/s HBM roofline, and the CPU oracle's rate on a sample."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import _lib, window
from pyorc_amd.project import Projection
from pyorc_amd.synth import projection_maps
lib = _lib.load(); _lib.require_device()

src = (2160, 3840) if os.environ.get("PROJ_4K") else (1080, 1920)
dst, T = (1080, 1920), int(sys.argv[1]) if len(sys.argv) > 1 else 401
idx_img, mask, src_idx, uidx, norm_idx = projection_maps(src, dst, tilt=0.3, seed=1)
if os.environ.get("PROJ_IDENTITY"):   # upper bound of the kernel structure: a gather that is a straight copy
    idx_img = mask = np.arange(dst[0] * dst[1], dtype=np.int64); src_idx = uidx = norm_idx = np.zeros(0, np.int64)
print(f"maps: nn {len(idx_img)} cells, groups {len(uidx)} with {len(src_idx)} samples (max {np.bincount(norm_idx).max() if len(norm_idx) else 0})")
p = Projection(src, dst, idx_img, mask, src_idx, uidx, norm_idx)
d_cam, d_ortho, d_out = C.c_void_p(), C.c_void_p(), C.c_void_p()
n_src, n_dst = src[0] * src[1], dst[0] * dst[1]
_lib.check(lib.lspiv_dev_malloc(C.byref(d_cam), T * n_src))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_ortho), T * n_dst * 4))
_lib.check(lib.lspiv_synth_particles_dev(d_cam, T, src[0], src[1], 3, 0.03))
nr, nc = window.get_array_shape(dst, (32, 32), (16, 16))
_lib.check(lib.lspiv_dev_malloc(C.byref(d_out), 16 * (T - 1) * nr * nc))

def proj(): p.project_frames_dev(d_cam.value, np.uint8, T, d_ortho.value)
def piv(): _lib.check(lib.lspiv_piv_pairs_dev(d_ortho, 1, T, dst[0], dst[1], 32, 32, 16, 16, -1.0, d_out, None, None))
def timed(fn, reps=5):
    fn(); _lib.check(lib.lspiv_synchronize())
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    _lib.check(lib.lspiv_synchronize())
    return (time.perf_counter() - t0) / reps

tp = timed(proj)
b_alg = n_src * 1 + n_dst * 4
print(f"project: {tp*1e3:.2f} ms / {T} frames -> {T/tp:.0f} frames/s, {T*b_alg/tp/1e9:.0f} GB/s algorithmic = {T*b_alg/tp/8e12*100:.1f}% of 8 TB/s")
tv = timed(piv, 3)
print(f"piv (float32 ortho): {tv*1e3:.2f} ms / {T-1} pairs -> {(T-1)/tv:.0f} pairs/s")
tc = timed(lambda: (proj(), piv()), 3)
print(f"chain project+piv: {tc*1e3:.2f} ms -> {(T-1)/tc:.0f} pairs/s from raw uint8 camera frames resident in HBM")
# parity + CPU rate on a sample
from oracle import project_oracle as pro
cam = np.empty((4,) + src, np.uint8); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(cam), d_cam, cam.nbytes))
got = np.empty((4,) + dst, np.float32); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(got), d_ortho, got.nbytes))
t0 = time.perf_counter(); ref = pro.project_frames(cam, dst, idx_img, mask, src_idx, uidx, norm_idx); t_cpu = (time.perf_counter() - t0) / 4
print(f"bit-exact vs oracle on 4 frames: {np.array_equal(got.astype(np.float64), ref)}; numpy oracle {1/t_cpu:.1f} frames/s (1 core)")
# a nearest-neighbour-only plan (reducer other than "mean"): float32 ortho stack vs the uint8-staying one
pn = Projection(src, dst, idx_img, mask)
def projn(): pn.project_frames_dev(d_cam.value, np.uint8, T, d_ortho.value)
def proj8(): pn.project_frames_dev(d_cam.value, np.uint8, T, d_ortho.value, keep_uint8=True)
def piv8(): _lib.check(lib.lspiv_piv_pairs_dev(d_ortho, 0, T, dst[0], dst[1], 32, 32, 16, 16, -1.0, d_out, None, None))
tn, tcn = timed(projn), timed(lambda: (projn(), piv()), 3)
t8, tc8 = timed(proj8), timed(lambda: (proj8(), piv8()), 3)
print(f"nearest-only plan, float32 ortho: project {tn*1e3:.2f} ms ({T/tn:.0f} frames/s), project+piv {(T-1)/tcn:.0f} pairs/s")
print(f"nearest-only plan, uint8 ortho:   project {t8*1e3:.2f} ms ({T/t8:.0f} frames/s, {T*(n_src+n_dst)/t8/1e9:.0f} GB/s algorithmic), project+piv {(T-1)/tc8:.0f} pairs/s")
got8 = np.empty((4,) + dst, np.uint8); _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(got8), d_ortho, got8.nbytes))
print(f"uint8 ortho equals the oracle on 4 frames: {np.array_equal(got8.astype(np.float64), pro.project_frames(cam, dst, idx_img, mask))}")
