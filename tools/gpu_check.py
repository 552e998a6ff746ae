"""Ad-hoc GPU-vs-oracle comparison used while developing kernels (not part of the test suite)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyorc_amd
from oracle import piv_oracle as po
from pyorc_amd.synth import particle_stack


def compare(frames, ws, ov, thr=None, tag=""):
    t0 = time.time()
    u, v, cm, sn, planes = pyorc_amd.piv_pairs(frames, ws, ov, thr, return_planes=True)
    t1 = time.time()
    nr, nc = u.shape[1:]
    x, y, corr = po.cross_corr(frames, ws, ov, signal_threshold=thr)
    uo, vo, cmo, sno = po.get_uv_timestep(frames, nc, nr, ws, ov, thr)
    print(f"[{tag}] shape {frames.shape} {frames.dtype} ws {ws} ov {ov} gpu {t1-t0:.3f}s")
    print("   planes max abs diff", np.nanmax(np.abs(planes - corr)), "nan mismatch", (np.isnan(planes) != np.isnan(corr)).sum())
    for g, r, n in ((u, uo, "u"), (v, vo, "v"), (cm, cmo, "cmax"), (sn, sno, "s2n")):
        nanm = (np.isnan(g) != np.isnan(r)).sum()
        with np.errstate(all="ignore"):
            err = np.nanmax(np.abs(g - r) / np.maximum(np.abs(r), 0.05)) if np.isfinite(r).any() else 0.0
        print(f"   {n}: rel err {err:.3e}  nan mismatch {nanm}  nan frac {np.isnan(r).mean():.3f}")


if __name__ == "__main__":
    fr = particle_stack(4, 160, 224, seed=3)
    compare(fr, (32, 32), (16, 16), tag="u8 32/16")
    compare(fr.astype(np.float32) - 20.0, (32, 32), (16, 16), tag="f32 signed")
    compare(fr.astype(np.float64), (32, 32), (24, 8), tag="f64 ov 24/8")
    compare(fr[:, :157, :211], (32, 32), (16, 16), tag="odd W")
    compare(fr, (32, 32), (16, 16), thr=0.35, tag="thr")
    compare(fr[:, :96, :96], (10, 10), (5, 5), tag="direct 10")
    compare(fr[:, :96, :128], (24, 16), (12, 8), tag="direct 24x16")
    compare(fr[:, :128, :128], (64, 64), (48, 48), tag="direct 64")
    z = fr.copy(); z[:, :40, :40] = 7
    compare(z, (32, 32), (16, 16), tag="const tile")
