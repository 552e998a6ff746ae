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
        # Round 3: the float64 rescue pass covers the ill-conditioned windows, so the 1e-4 gate runs over ALL windows.  The
        # only ones set aside are EXACT float64 ties of the plane maximum (relative gap < 1e-12 on a non-zero plane): which
        # of two equal samples is "the" arg-max is decided by the rounding of whatever transform computed them -- the
        # oracle's own answer there is not reproducible by any other implementation, ffpiv's FFT included.
        # (a plane below 1e-12 is zero in exact arithmetic -- the normalised windows are >= 0 --: all its samples tie)
        tie = ((cond[..., 0] < 1e-12) | (cm < 1e-12)) & (cm > 0)
        ok = ~tie
        well = c_oracle.well_posed(cond)
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
        out["parity_windows_ill_posed"] = 0   # round 2 excluded the ill-conditioned windows here; they are gated now
        out["parity_windows_float64_ties"] = int(tie.sum())
        out["parity_float64_ties"] = ties_report(gpu_block, (u, v, cm, sn), tie)
        out["parity_ill_conditioned_now_gated"] = int((~well & ok).sum())
    return out


def parity_only(frames_sample: np.ndarray, ws, ov, gpu_block) -> dict:
    """The gate of :func:`run` without the timing: the product's (4, P, n_rows, n_cols) block against the C oracle on EVERY window
    of the sample's P pairs (exact float64 ties of the plane maximum set aside and counted).  bench.py runs it for every single-GPU
    BASELINE configuration on a bounded number of pairs, so that each driver run carries a full-grid oracle check of all of them."""
    cores = effective_cores()
    t0 = time.perf_counter()
    u, v, cm, sn, cond = c_oracle.piv_pairs(frames_sample, ws, ov, nthreads=cores, return_cond=True)
    dt = time.perf_counter() - t0
    tie = ((cond[..., 0] < 1e-12) | (cm < 1e-12)) & (cm > 0)
    ok = ~tie
    worst, nan_mismatch = 0.0, 0
    for g, r in zip(gpu_block, (u, v, cm, sn)):
        nan_mismatch += int((np.isnan(g) != np.isnan(r))[ok].sum())
        with np.errstate(all="ignore"):
            e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[ok]
        if np.isfinite(e).any():
            worst = max(worst, float(np.nanmax(e)))
    return {"pairs": int(frames_sample.shape[0] - 1), "windows_checked": int(ok.sum()), "exact_ties_set_aside": int(tie.sum()),
            "max_rel_err_vs_oracle": float(f"{worst:.3e}"), "nan_mismatch": nan_mismatch, "gate": 1e-4,
            "passed": bool(worst <= 1e-4 and nan_mismatch == 0), "oracle_s": round(dt, 2)}


def ties_report(gpu_block, ref, tie) -> dict:
    """The windows the gate sets aside: exact float64 ties of the plane maximum, listed (first 16) with both answers."""
    idx = np.argwhere(tie)
    rep = {"windows": int(tie.sum())}
    if idx.size:
        gu, gv = gpu_block[0], gpu_block[1]
        ru, rv = ref[0], ref[1]
        same = int(sum(1 for i in map(tuple, idx)
                       if np.allclose([gu[i], gv[i]], [ru[i], rv[i]], rtol=0, atol=1e-4, equal_nan=True)))
        rep["same_answer_anyway"] = same
        rep["first"] = [{"pair_row_col": [int(k) for k in i], "gpu_uv": [float(gu[i]), float(gv[i])],
                         "oracle_uv": [float(ru[i]), float(rv[i])], "corr": float(ref[2][i])} for i in map(tuple, idx[:16])]
    return rep


def ensemble_parity(frames_sample: np.ndarray, ws, ov, got_u, got_v, corr_min: float, s2n_min: float, count_min: float) -> dict:
    """Ensemble mode (pyorc/velocimetry/ffpiv.py:182-376) of the numpy oracle on a few frames of the benchmark stack against the
    product's u, v (1, n_rows, n_cols) for the same frames: every window but exact ties of the oracle's mean-plane maximum."""
    from . import piv_oracle as po

    T = frames_sample.shape[0]
    t0 = time.perf_counter()
    ref = po.get_ffpiv(frames_sample, np.ones(T - 1), tuple(ws), tuple(ov), 1.0, 1.0, ensemble_corr=True, corr_min=corr_min,
                       s2n_min=s2n_min, count_min=count_min)
    dt = time.perf_counter() - t0
    cmean = np.asarray(ref["corr_mean"], dtype=np.float64).reshape(-1, ws[0] * ws[1])
    with np.errstate(all="ignore"):
        top2 = np.sort(np.nan_to_num(cmean, nan=-1.0), axis=1)[:, -2:]
        tie = ((top2[:, 1] > 0) & (top2[:, 0] >= top2[:, 1] * (1.0 - 1e-9))).reshape(ref["v_x"].shape[1:])
    worst, nan_bad = 0.0, 0
    for g, k in ((got_u, "v_x"), (got_v, "v_y")):
        g, r = np.asarray(g, dtype=np.float64).reshape(tie.shape), np.asarray(ref[k], dtype=np.float64)[0]
        nan_bad += int((np.isnan(g) != np.isnan(r))[~tie].sum())
        with np.errstate(all="ignore"):
            e = (np.abs(g - r) / np.maximum(np.abs(r), 0.05))[~tie]
        if np.isfinite(e).any():
            worst = max(worst, float(np.nanmax(e)))
    return {"window": list(ws), "overlap": list(ov), "pairs": T - 1, "windows_checked": int((~tie).sum()), "exact_ties_set_aside": int(tie.sum()),
            "max_rel_err_vs_oracle": float(f"{worst:.3e}"), "nan_mismatch": nan_bad, "oracle_s": round(dt, 2)}


# ---- numpy / pocketfft reading of the same sample (promised beside the C port) -----------------------------
_NP_SAMPLE = None      # (frames, ws, ov) inherited by the forked workers


def _np_pairs(span):
    from . import piv_oracle as po

    frames, ws, ov = _NP_SAMPLE
    a, b = span
    n_cols = po.get_axis_shape(frames.shape[2], ws[1], ov[1])
    n_rows = po.get_axis_shape(frames.shape[1], ws[0], ov[0])
    u, v, cm, sn = po.get_uv_timestep(frames[a:b + 1], n_cols, n_rows, tuple(ws), tuple(ov))
    return a, np.asarray(u, np.float64), np.asarray(v, np.float64)


def numpy_pocketfft(frames_sample: np.ndarray, ws, ov, pairs_per_core: int = 2, check=None) -> dict:
    """The numpy oracle's path -- ``sliding_window_stack`` + batched ``numpy.fft.rfft2`` / ``irfft2`` (pocketfft, the FFT family
    rocket_fft wraps) + numpy reductions, float64: the reference's own data flow, window stack and correlation volume included -- on
    the first pairs of the same sample, one PROCESS per core (``multiprocessing``, fork), ``pairs_per_core`` pairs each.  Reported
    beside the C port; ``check``: (u, v) of the C port for those pairs, to state the agreement of the two restatements."""
    import multiprocessing as mp

    global _NP_SAMPLE
    cores = effective_cores()
    n_pairs = int(min(frames_sample.shape[0] - 1, max(1, pairs_per_core) * cores))
    spans = [(p, p + 1) for p in range(n_pairs)]
    _NP_SAMPLE = (frames_sample[:n_pairs + 1], tuple(ws), tuple(ov))
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(processes=cores) as pool:
            pool.map(_np_pairs, spans[:cores])          # warm-up: imports, page faults, one pair per worker
            t0 = time.perf_counter()
            res = pool.map(_np_pairs, spans, chunksize=1)
            dt = time.perf_counter() - t0
    finally:
        _NP_SAMPLE = None
    out = {"value": round(n_pairs / dt, 3), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
           "sample": f"first {n_pairs} frame-pairs of the same sample, numpy oracle (oracle/piv_oracle.py: window stack + batched numpy.fft.rfft2 / "
                     f"irfft2 = pocketfft, float64, numpy reductions), {cores} processes x 1 thread, {dt:.1f} s wall"}
    if check is not None:
        worst = 0.0
        for a, u, v in res:
            for g, r in ((u[0], check[0][a]), (v[0], check[1][a])):
                with np.errstate(all="ignore"):
                    e = np.abs(g - r) / np.maximum(np.abs(r), 0.05)
                if np.isfinite(e).any():
                    worst = max(worst, float(np.nanmax(e[np.isfinite(e)])))
        out["max_rel_diff_vs_c_port"] = float(f"{worst:.3e}")
    return out
