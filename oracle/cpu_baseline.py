"""CPU-baseline leg of bench.py: times the C oracle (all host cores) on a bounded sample.  TEST INFRASTRUCTURE.

The figure is "own restatement of the ffpiv semantics" (kind = "port"), NOT ffpiv itself: neither ffpiv,
rocket_fft nor numba exist on the GPU box, and nothing of /root/reference travels there.
"""

from __future__ import annotations

import os
import time

import numpy as np

from . import c_oracle


def effective_cores() -> int:
    """Host cores this process may really use: min(online CPUs, affinity mask, cgroup CPU quota).

    The GPU boxes expose 256 logical CPUs but cap the container at 16 via cgroup cpu.max; running 256 OpenMP
    threads there is 6x SLOWER than 16, so the baseline uses the quota and reports it as ``cores``.
    """
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, min(n, c_oracle.max_threads()))


def default_sample_pairs(H: int, W: int, ws) -> int:
    """About 10-30 s of CPU work: ~20-50 us per 32x32 window pair per core (measured), scaled by area."""
    cores = effective_cores()
    n_win = ((H - ws[0]) // (ws[0] // 2) + 1) * ((W - ws[1]) // (ws[1] // 2) + 1)
    per_pair_s = n_win * 30e-6 * (ws[0] * ws[1] / 1024.0) ** 1.2 / cores
    return int(max(2, min(1000, round(15.0 / max(per_pair_s, 1e-6)))))


def run(frames_sample: np.ndarray, ws, ov, gpu_block=None) -> dict:
    """Time the oracle on ``frames_sample`` (T,H,W); compare with the GPU block (4, T-1, n_rows, n_cols)."""
    cores = effective_cores()
    n_pairs = frames_sample.shape[0] - 1
    c_oracle.piv_pairs(frames_sample[:2], ws, ov, nthreads=cores)  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    u, v, cm, sn = c_oracle.piv_pairs(frames_sample, ws, ov, nthreads=cores)
    dt = time.perf_counter() - t0
    if gpu_block is not None:  # untimed: grade the windows (argmax gap, neighbourhood floor) for the parity line
        *_, cond = c_oracle.piv_pairs(frames_sample, ws, ov, nthreads=cores, return_cond=True)
        ok = c_oracle.well_posed(cond)
    out = {
        "value": round(n_pairs / dt, 3),
        "unit": "frame-pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_pairs} frame-pairs of the benchmark stack ({frames_sample.shape[1]}x"
                  f"{frames_sample.shape[2]}, {ws[0]}x{ws[1]} windows), C/OpenMP float64 restatement of the ffpiv "
                  f"semantics (oracle/piv_oracle.c), {dt:.1f} s wall",
    }
    if gpu_block is not None:
        worst = 0.0
        nan_mismatch = 0
        for g, r in zip(gpu_block, (u, v, cm, sn)):
            nan_mismatch += int((np.isnan(g) != np.isnan(r))[ok].sum())
            with np.errstate(all="ignore"):
                e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[ok]
            if np.isfinite(e).any():
                worst = max(worst, float(np.nanmax(e)))
        out["parity_max_rel_err_vs_oracle"] = float(f"{worst:.3e}")
        out["parity_nan_mismatch"] = nan_mismatch
        out["parity_windows_checked"] = int(ok.sum())
        out["parity_windows_ill_posed"] = int((~ok).sum())
        out["parity_ill_posed"] = ill_posed_report(gpu_block, (u, v, cm, sn), ok)
    return out


def ill_posed_report(gpu_block, ref, ok) -> dict:
    """What the 1e-4 gate does NOT cover: the windows `c_oracle.well_posed` excludes (arg-max gap < 1e-5, a peak
    neighbour < 2 % of the maximum, log-curvature < 0.05), graded on their own -- NaN-mask mismatches, the error
    distribution of u, v and of corr / s2n, and how many of them a pyorc user would ever see: the share that passes
    the default post-PIV masks `mask.corr(tolerance=0.1)` and `mask.s2n(tolerance=10)` (pyorc/api/mask.py:204,216)."""
    bad = ~ok
    n = int(bad.sum())
    rep = {"windows": n}
    if n == 0:
        return rep
    gu, gv, gc, gs = (np.asarray(a, dtype=np.float64) for a in gpu_block)
    ru, rv, rc, rs = (np.asarray(a, dtype=np.float64) for a in ref)
    rep["nan_mismatch_uv"] = int(((np.isnan(gu) != np.isnan(ru)) | (np.isnan(gv) != np.isnan(rv)))[bad].sum())
    rep["nan_mismatch_corr_s2n"] = int(((np.isnan(gc) != np.isnan(rc)) | (np.isnan(gs) != np.isnan(rs)))[bad].sum())
    with np.errstate(all="ignore"):
        for name, g, r in (("u", gu, ru), ("v", gv, rv), ("corr", gc, rc), ("s2n", gs, rs)):
            e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[bad]
            e = e[np.isfinite(e)]
            if e.size:
                rep[f"rel_err_{name}"] = {"p50": float(f"{np.percentile(e, 50):.3e}"), "p99": float(f"{np.percentile(e, 99):.3e}"),
                                          "max": float(f"{e.max():.3e}"), "n": int(e.size)}
        survive = bad & (rc >= 0.1) & (rs >= 10.0) & np.isfinite(ru) & np.isfinite(rv)   # NaN compares false
        rep["survive_default_corr_s2n_masks"] = int(survive.sum())
        rep["survive_share_of_ill_posed"] = float(f"{survive.sum() / n:.4f}")
        if survive.any():
            e = np.maximum(np.abs(gu - ru), np.abs(gv - rv))[survive]   # absolute, pixels: what reaches a velocity field
            e = e[np.isfinite(e)]
            if e.size:
                rep["surviving_abs_err_px"] = {"p50": float(f"{np.percentile(e, 50):.3e}"), "p99": float(f"{np.percentile(e, 99):.3e}"),
                                               "max": float(f"{e.max():.3e}")}
            rep["surviving_nan_mismatch"] = int((np.isnan(gu) | np.isnan(gv))[survive].sum())
    return rep
