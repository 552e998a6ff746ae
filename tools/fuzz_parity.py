"""Randomised differential test: the HIP path vs the C oracle over random shapes / dtypes / windows / overlaps /
thresholds.  Prints one line per case and a summary; exits non-zero on any gate violation.  FUZZ_WIDE=1: wide window grids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyorc_amd
from oracle import c_oracle
from pyorc_amd.synth import particle_stack

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t_start = time.time()
for case in range(n_cases):
    ws = int(rng.choice([32, 32, 32, 64, 64, 16, 24, 10, 48, 8, 6, 12, 20, 14, 18, 22, 26, 28, 30, 36, 40, 44, 50, 56, 60, 62, 9, 25, 35,
                         72, 80, 96, 100, 112, 128, 66, 97,     # above 64: four-step sizes, a 2 x odd and an odd one (DFT passes)
                         17, 33, 41, 49, 63,                    # odd: embedded below 33, direct up to 39, DFT passes from 41
                         74, 98, 110, 114, 125, 126,            # composite lengths in two DFT passes (126, 114: in two rounds)
                         144, 160, 200]))                       # above 128: the same passes on HBM slots
    wsy = ws if rng.random() < 0.8 else int(rng.choice([8, 16, 20, 32, 80, 136]))
    ov = (int(rng.integers(0, wsy)), int(rng.integers(0, ws)))
    H = int(rng.integers(wsy, wsy * 4 + 7)); W = int(rng.integers(ws, ws * 5 + 9)); T = int(rng.integers(2, 6)) if rng.random() < 0.6 else int(rng.integers(6, 14)) if rng.random() < 0.8 or ws > 40 else int(rng.integers(26, 60))   # sometimes across a 25-pair anchor
    if os.environ.get("FUZZ_WIDE") and ws <= 64:   # grids of 26 ... 70 columns: wider than the walking kernels' job strips (24 / 32 windows)
        W = ws + max(ws - ov[1], 1) * int(rng.integers(25, 70)) + int(rng.integers(0, 5)); H = int(rng.integers(wsy, 2 * wsy + 9))
    dtype = rng.choice([np.uint8, np.float32, np.float64])
    thr = None if rng.random() < 0.6 else float(rng.uniform(0, 0.6))
    fr = particle_stack(T, H, W, seed=int(rng.integers(1 << 30)), density=float(rng.uniform(0.01, 0.08)))
    if dtype != np.uint8:
        fr = (fr.astype(dtype) * float(rng.uniform(0.1, 3)) - float(rng.uniform(0, 50)))
    if rng.random() < 0.3:
        fr[:, : H // 3, : W // 3] = 7 if dtype == np.uint8 else float(rng.choice([0.0, 7.0, -2.5, 0.1]))  # masked area
    if rng.random() < 0.2:
        fr[rng.integers(T)] = 0                           # an empty frame
    u, v, cm, sn = pyorc_amd.piv_pairs(fr, (wsy, ws), ov, thr)
    uo, vo, cmo, sno, cond = c_oracle.piv_pairs(fr, (wsy, ws), ov, thr, return_cond=True)
    ok = ~c_oracle.exact_tie(cond, cmo)   # round 3: every window but exact float64 ties (the float64 rescue pass covers the ill-conditioned ones)
    uniq = ok
    def err(g, r, m=None):
        with np.errstate(all="ignore"):
            e = np.abs(g - r) / np.maximum(np.abs(r), 0.05)
        e = e if m is None else e[m]
        return float(np.nanmax(e)) if np.isfinite(e).any() else 0.0
    nan_bad = int((np.isnan(cm) != np.isnan(cmo)).sum() + (np.isnan(sn) != np.isnan(sno)).sum() + (np.isnan(u) != np.isnan(uo))[uniq].sum())
    e_c, e_s, e_u, e_v = err(cm, cmo), err(sn, sno), err(u, uo, ok), err(v, vo, ok)
    fail = nan_bad > 0 or max(e_c, e_s, e_u, e_v) > 1e-4
    bad += fail
    if fail and os.environ.get("FUZZ_DUMP"):
        os.makedirs(os.environ["FUZZ_DUMP"], exist_ok=True)
        np.savez_compressed(os.path.join(os.environ["FUZZ_DUMP"], f"case{case}.npz"), fr=fr, ws=(wsy, ws), ov=ov, thr=-1 if thr is None else thr,
                            u=u, v=v, cm=cm, sn=sn, uo=uo, vo=vo, cmo=cmo, sno=sno)
    print(f"{'FAIL' if fail else 'ok  '} {case:3d} win ({wsy},{ws}) ov {ov} frame ({T},{H},{W}) {np.dtype(dtype).name:7s} thr {thr} "
          f"gated {ok.mean():.3f} errs c {e_c:.1e} s {e_s:.1e} u {e_u:.1e} v {e_v:.1e} nan {nan_bad}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t_start:.1f} s")
sys.exit(1 if bad else 0)
