"""Where does ensemble mode lose the 1e-4 gate?  GPU ensemble (get_piv, ensemble_corr=True) against the oracle's get_ffpiv on a
few shapes; for every window above the gate: the error, the oracle's peak neighbourhood on its float64 mean plane (conditioning)
and the GPU's float32 mean plane at the same five samples (plane noise).  usage: ens_diag.py [tol]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import piv_oracle as po
from pyorc_amd import piv
from pyorc_amd.synth import particle_stack

warnings.simplefilter("ignore")
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
CASES = [(32, 3, 128, 160, 0.03), (32, 9, 128, 160, 0.03), (32, 29, 160, 224, 0.03), (32, 101, 160, 224, 0.02), (32, 401, 128, 160, 0.02),
         (64, 5, 256, 256, 0.05), (64, 29, 200, 280, 0.03), (64, 201, 200, 280, 0.02), (24, 6, 96, 96, 0.05), (16, 20, 96, 128, 0.05),
         (96, 6, 212, 288, 0.03), (128, 6, 276, 384, 0.03)]
for ws, T, H, W, dens in CASES:
    fr = particle_stack(T, H, W, seed=ws + T, density=dens)
    kw = dict(corr_min=0.1, s2n_min=1.5)
    ens = piv.Ensemble((H, W), (ws, ws), (ws // 2, ws // 2))
    ens.accumulate(fr, kw["corr_min"], kw["s2n_min"])
    u, v, cnt, mean = ens.finish(0.0, 1, return_mean=True)
    ens.close()
    ref = po.get_ffpiv(fr, np.ones(T - 1), (ws, ws), (ws // 2, ws // 2), 1.0, 1.0, ensemble_corr=True, count_min=0.0, **kw)
    rm = np.asarray(ref["corr_mean"]).reshape(-1, ws, ws)
    gm = mean.reshape(-1, ws, ws).astype(np.float64)
    uo, vo = ref["v_x"][0].astype(np.float64).ravel(), ref["v_y"][0].astype(np.float64).ravel()
    ug, vg = u[0].astype(np.float64).ravel(), v[0].astype(np.float64).ravel()
    with np.errstate(all="ignore"):
        eu = np.abs(ug - uo) / np.maximum(np.abs(uo), 0.05)
        ev = np.abs(vg - vo) / np.maximum(np.abs(vo), 0.05)
    e = np.fmax(eu, ev)
    nanbad = int((np.isnan(ug) != np.isnan(uo)).sum())
    with np.errstate(all="ignore"):
        pn = np.nanmax(np.abs(gm - rm), axis=(1, 2)) / np.nanmax(rm, axis=(1, 2))      # plane noise relative to the plane maximum
    bad = np.where(e > tol)[0]
    print(f"== win {ws} T {T} frame {H}x{W}: {e.size} windows, nan mismatches {nanbad}, above {tol:g}: {bad.size}, max err {np.nanmax(e):.2e}, "
          f"plane noise / max: median {np.nanmedian(pn):.1e} worst {np.nanmax(pn):.1e}, count median {np.median(cnt):.0f}", flush=True)
    for w in bad[:6]:
        p = rm[w]
        i, j = np.unravel_index(np.nanargmax(p), p.shape)
        if 0 < i < ws - 1 and 0 < j < ws - 1:
            nb = [p[i, j], p[i - 1, j], p[i + 1, j], p[i, j - 1], p[i, j + 1]]
            gb = [gm[w][i, j], gm[w][i - 1, j], gm[w][i + 1, j], gm[w][i, j - 1], gm[w][i, j + 1]]
            srt = np.sort(p.ravel())
            print(f"   w {w}: err {e[w]:.2e} u {ug[w]:.5f}/{uo[w]:.5f} v {vg[w]:.5f}/{vo[w]:.5f} count {cnt[w]:.0f} peak ({i},{j}) ref5 "
                  + " ".join(f"{x:.3e}" for x in nb) + " | gpu-ref " + " ".join(f"{a - b:+.1e}" for a, b in zip(gb, nb))
                  + f" | runner-up gap {(srt[-1] - srt[-2]) / srt[-1]:.1e}")
        else:
            print(f"   w {w}: err {e[w]:.2e} border peak ({i},{j}) u {ug[w]}/{uo[w]}")
