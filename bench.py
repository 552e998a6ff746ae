#!/usr/bin/env python
"""Headline benchmark: PIV frame-pairs/s on a synthetic 1080p stack, 32x32 windows @ 50 % overlap.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (lspiv_piv_pairs_dev: window gather + normalise + FFT
cross-correlation + corr_max / s2n + sub-pixel peak, fused) over ONE batch of 1000 frame pairs
per GPU that is already resident in HBM (BASELINE.json configs[1]).  With N > 1 every rank owns
its own 1000-pair time block (BASELINE.json configs[4]: 8 x 1000 pairs on 8 GPUs; per-GPU work is fixed, so the
line says "weak", and north_star's ">= 6.5x at 8 GPUs" is value(N=8) / value(N=1) of this very command); the only
exchange is the RCCL all-gather of the packed (4, t, y, x) result block, software-pipelined one step behind the kernel.

No PyTorch anywhere: kernels, streams, events and the RCCL exchange all go through the C ABI (include/lspiv.h).
`python bench.py --gpus N` launches its N ranks itself; under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it runs as one of them.  Either way rank 0
prints ONE JSON line (see the task contract).

At N = 1 the line also carries, measured after the timed region: the roofline of the dominant kernel, the CPU baseline
(C oracle on a bounded sample, with the live parity check incl. the ill-posed windows), the other two single-GPU
BASELINE configs (`config.other_configs`: C3 64x64 @ 75 %, C4 4K) and the host-fed (PCIe-inclusive) rates.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pyorc_amd import _lib, window  # noqa: E402
from pyorc_amd import comm as _comm_mod  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)   # the first ~5 launches after an idle gap run up to 25 % slow (tools/clock_ramp.py)
    ap.add_argument("--pairs", type=int, default=1000, help="frame pairs per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--window", type=int, default=32)
    ap.add_argument("--overlap", type=int, default=16)
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs of the CPU-baseline sample (-1: auto, 0: skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other-configs and host-fed legs (N = 1)")
    ap.add_argument("--sustained-s", type=float, default=10.0,
                    help="N = 1: seconds of the sustained loop run after the timed steps (clock and socket power sampled; 0: skip)")
    ap.add_argument("--parity-pairs", type=int, default=50, help="pairs of the full-grid oracle check of configs[2] / configs[3] (0: skip)")
    ap.add_argument("--seed", type=int, default=20260927 + 2)
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: cut a FIXED total of --strong-pairs over the N ranks (north_star's 'strong scaling', literally) "
                         "instead of --pairs per rank (the default, 'weak')")
    ap.add_argument("--strong-pairs", type=int, default=8000, help="total frame pairs of --strong (BASELINE.json configs[4]: 8000)")
    return ap.parse_args()


def flop_per_pair(ws: int, n_win: int) -> float:
    """per window 3 real 2-D FFTs at 2.5 N log2 N (N = ws^2) + 6 ws (ws/2 + 1) for the cross
    spectrum + 4 N for the normalisation."""
    n = ws * ws
    return n_win * (3 * 2.5 * n * np.log2(n) + 6 * ws * (ws // 2 + 1) + 4 * n)


def kernel_name_for(ws: int, pairs: int) -> str:
    walking = os.environ.get("LSPIV_WALK", "1") != "0" and ws % 2 == 0 and 6 <= ws <= 64
    return f"piv_fft_{'walk_' if walking else ''}kernel<unsigned char, {ws}, false, false>"


def measured_traffic(kernel_substr: str, pairs: int, H: int, W: int, window_size: int, overlap: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_summary.json,
    written by tools/summarize_profile.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this very
    command).  Only returned when the profiled launch had the same shape and kernel AND the summary carries the hash of the
    kernel sources in this tree (`code_hash`, pyorc_amd._lib.kernel_code_hash): change the kernel without re-profiling
    and the line says `traffic: null`, not a stale number.  Latest matching summary wins."""
    import glob

    best = None
    code = _lib.kernel_code_hash()
    if _lib.binary_provenance()["binary_kernel_hash"] != code:   # the loaded binary was not built from these kernel sources (LSPIV_ALLOW_STALE / LSPIV_LIBRARY)
        return None
    want = {"pairs": pairs, "H": H, "W": W, "window": window_size, "overlap": overlap}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        launch = d.get("launch", {"pairs": 1000, "H": 1080, "W": 1920, "window": 32, "overlap": 16})
        if launch != want or d.get("code_hash") != code:   # another shape, or profiled on other kernel code: not this kernel's traffic
            continue
        for name, k in d.get("kernels", {}).items():
            if kernel_substr in name and "hbm_traffic_bytes" in k:
                best = {"bytes": round(k["hbm_traffic_bytes"]), "source": os.path.basename(f)}
    return best


def roofline_block(lib, kernel_ms: float, pairs: int, H: int, W: int, ws: int, ov: int, n_win: int, launch_ms=None) -> dict:
    b_alg_pair = 2 * H * W * 1 + 16 * n_win  # both frames read once + 4 f32 per window
    achieved = b_alg_pair * pairs / (kernel_ms * 1e-3) / 1e9
    fpp = flop_per_pair(ws, n_win)
    name = kernel_name_for(ws, pairs)
    r = {
        "bound": "hbm",
        "kernel": name,
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": None,
        "algorithmic_bytes_per_launch": b_alg_pair * pairs,
        "algorithmic_bytes_per_pair": b_alg_pair,
        "kernel_ms_per_launch": round(kernel_ms, 4),
        "note": "FFT path is FP32-VALU/LDS bound, not HBM bound (DESIGN.md section 4); secondary bound below",
        "secondary": {"bound": "fp32-valu", "flop_per_pair": round(fpp),
                      "achieved_tflops": round(fpp * pairs / (kernel_ms * 1e-3) / 1e12, 2), "peak_tflops": FP32_VALU_TFLOPS,
                      "frac": round(fpp * pairs / (kernel_ms * 1e-3) / 1e12 / FP32_VALU_TFLOPS, 4)},
    }
    tr = measured_traffic(name.split(",")[0] + "," + name.split(",")[1] + ",", pairs, H, W, ws, ov)
    r["kernel_code_hash"] = _lib.kernel_code_hash()
    if launch_ms is not None:
        r["launch_ms_with_rescue_kernels"] = round(launch_ms, 4)
    if tr:
        r["traffic"] = tr["bytes"]
        r["traffic_source"] = f"profiles/{tr['source']} (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
    return r


def time_launches(lib, launch, reps: int) -> float:
    """Average launch duration in ms, HIP events on the library's launch stream."""
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_event_create(C.byref(ev0)))
    _lib.check(lib.lspiv_event_create(C.byref(ev1)))
    _lib.check(lib.lspiv_synchronize())
    _lib.check(lib.lspiv_event_record(ev0))
    for _ in range(reps):
        launch()
    _lib.check(lib.lspiv_event_record(ev1))
    ms = C.c_float()
    _lib.check(lib.lspiv_event_elapsed_ms(ev0, ev1, C.byref(ms)))
    _lib.check(lib.lspiv_event_destroy(ev0))
    _lib.check(lib.lspiv_event_destroy(ev1))
    return ms.value / reps


def time_kernel_only(lib, launch, reps: int):
    """Mean duration [ms] of the PIV kernel inside `reps` launches issued the way a caller issues them (rescue pass ON), by the
    library's own HIP events right around that kernel on the launch stream (include/lspiv.h, lspiv_kernel_times): the quantity
    `rocprofv3 --kernel-trace` reports per launch of the kernel -- the committed profiles/ summaries hold the same figure."""
    reps = min(reps, 16)
    _lib.check(lib.lspiv_synchronize())
    _lib.set_option("time_kernel", 1)
    try:
        buf, n = (C.c_float * 16)(), C.c_int(0)
        _lib.check(lib.lspiv_kernel_times(buf, 16, C.byref(n)))     # empty the ring
        for _ in range(reps):
            launch()
        _lib.check(lib.lspiv_synchronize())
        _lib.check(lib.lspiv_kernel_times(buf, 16, C.byref(n)))
    finally:
        _lib.set_option("time_kernel", 0)
    ms = [float(buf[i]) for i in range(n.value)]
    return (sum(ms) / len(ms) if ms else None), ms


class SmiSampler:
    """Shader clock [MHz] and socket power [W] of device 0 once a second while a measurement runs (rocm-smi in a thread;
    measurement code only).  Missing tool / unparsable output: no samples, the figures are None."""

    def __init__(self, period: float = 1.0):
        import threading

        self.period, self.samples, self._stop = period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re

        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                clk = re.search(r"sclk clock level:[^\n]*\((\d+)Mhz\)", txt)
                pw = re.search(r"Package Power \(W\):\s*([0-9.]+)", txt)
                if clk and pw:
                    self.samples.append((time.time(), int(clk.group(1)), float(pw.group(1))))
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(timeout=15)
        return False

    def summary(self, t0: float, t1: float) -> dict:
        inside = [(c, p) for (t, c, p) in self.samples if t0 + 1.0 <= t <= t1]     # (the first second is the clock ramp)
        if not inside:
            return {"samples": 0, "sclk_mhz_mean": None, "socket_power_w_mean": None}
        return {"samples": len(inside), "sclk_mhz_mean": round(float(np.mean([c for c, _ in inside])), 1),
                "sclk_mhz_min": int(min(c for c, _ in inside)), "socket_power_w_mean": round(float(np.mean([p for _, p in inside])), 1),
                "socket_power_w_max": round(max(p for _, p in inside), 1)}


def sustained_run(lib, step, sync, pairs_per_step: int, seconds: float) -> dict:
    """The step loop for `seconds` of wall time (>= 10 s by default): what a long job sees once the socket sits at its power limit --
    `value` times a 20-step burst of 0.12 s (eleven boxes, 151-162 k in bursts, 151 k sustained)."""
    sync()
    with SmiSampler() as smi:
        t0w = time.time()
        t0 = time.perf_counter()
        n = 0
        while True:
            for _ in range(50):
                step()
            n += 50
            sync()
            if time.perf_counter() - t0 >= seconds:
                break
        dt = time.perf_counter() - t0
        t1w = time.time()
    return {"seconds": round(dt, 3), "steps": n, "pairs_per_s": round(n * pairs_per_step / dt, 1), "ms_per_step": round(dt / n * 1e3, 4),
            **smi.summary(t0w, t1w)}


def other_config(lib, name: str, d_frames, pairs: int, H: int, W: int, ws: int, ov: int, reps: int = 3, parity_pairs: int = 0) -> dict:
    """One more BASELINE.json single-GPU configuration on an HBM-resident stack: kernel time by HIP events."""
    n_rows, n_cols = window.get_array_shape((H, W), (ws, ws), (ov, ov))
    n_win = n_rows * n_cols
    d_out = C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_out), 4 * pairs * n_win * 4))

    def go():
        _lib.check(lib.lspiv_piv_pairs_dev(d_frames, 0, pairs + 1, H, W, ws, ws, ov, ov, -1.0, d_out, None, None))

    go()
    go()                                              # two untimed launches (26-30 ms each): past the slow first launches after an idle gap
    _lib.check(lib.lspiv_synchronize())
    ms = time_launches(lib, go, reps)                 # the launch as a caller issues it: PIV kernel + rescue kernels
    st = (C.c_int64 * 5)()
    _lib.check(lib.lspiv_rescue_stats(None, st))
    kernel_ms, _ = time_kernel_only(lib, go, reps)    # the dominant kernel inside such launches (roofline; = the profile's per-kernel time)
    parity = None
    if parity_pairs > 0:
        # full-grid oracle check of this configuration on a bounded number of pairs (every driver run, every single-GPU config)
        from oracle import cpu_baseline as cb   # test infrastructure: parity leg only

        n_s = min(parity_pairs, pairs)
        sample = np.empty((n_s + 1, H, W), dtype=np.uint8)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(sample), d_frames, sample.nbytes))
        block = np.empty((4, n_s, n_rows, n_cols), dtype=np.float32)     # the first n_s pairs of each of [u | v | corr | s2n]
        for k in range(4):
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(block[k]), C.c_void_p(d_out.value + k * pairs * n_win * 4), block[k].nbytes))
        parity = cb.parity_only(sample, (ws, ws), (ov, ov), block)
        del sample, block
    _lib.check(lib.lspiv_dev_free(d_out))
    return {
        "workload": name,
        "pairs_per_s": round(pairs / (ms * 1e-3), 1),
        "mvectors_per_s": round(pairs * n_win / (ms * 1e-3) / 1e6, 2),
        "windows_per_pair": n_win,
        "kernel": kernel_name_for(ws, pairs),
        "launch_ms": round(ms, 4),
        "kernel_ms": round(kernel_ms, 4),
        "rescued_windows_per_launch": {"fit": int(st[0]), "amb": int(st[1])},
        **({"parity_vs_oracle": parity} if parity is not None else {}),
        "roofline": {k: v for k, v in roofline_block(lib, kernel_ms, pairs, H, W, ws, ov, n_win).items()
                     if k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "secondary",
                              "algorithmic_bytes_per_pair")},
    }


def ensemble_config(lib, name: str, d_frames, pairs: int, H: int, W: int, ws: int, ov: int, reps: int = 3) -> dict:
    """Ensemble-correlation mode (pyorc/velocimetry/ffpiv.py:182-376) on the HBM-resident stack: every pair's
    masked correlation plane is added to the running sum (walking ensemble kernel + the ordered merge of its per-segment partial
    sums), then one finish (mean planes, peak fit, float64 rescue of the flagged fits).  Launch time by HIP events."""
    from pyorc_amd import piv

    ens = piv.Ensemble((H, W), (ws, ws), (ov, ov))
    ens.set_retain(piv.Ensemble.RETAIN_BORROW)   # the stack stays in HBM until finish: the final fit can go back to the frames
    n_win = ens.n_rows * ens.n_cols
    d_cs = C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(d_cs), 8 * pairs * n_win))

    def go():
        ens.accumulate_dev(d_frames.value, np.uint8, pairs + 1, 0.2, 3.0, d_cs.value)

    go()
    go()
    _lib.check(lib.lspiv_synchronize())
    ms = time_launches(lib, go, reps)
    t0 = time.perf_counter()
    u, v, cnt = ens.finish(0.2, 1)
    fin_ms = (time.perf_counter() - t0) * 1e3
    st = ens.stats()
    ens.close()
    _lib.check(lib.lspiv_dev_free(d_cs))
    b_alg_pair = 2 * H * W + 8 * n_win           # both frames read once + masked corr_max, s2n per window (the plane sum stays in HBM)
    achieved = b_alg_pair * pairs / (ms * 1e-3) / 1e9
    kernel = f"piv_fft_walk_ensemble_kernel<unsigned char, {ws}, false>"
    tr = measured_traffic(kernel, pairs, H, W, ws, ov)
    return {
        "workload": name, "pairs_per_s": round(pairs / (ms * 1e-3), 1), "launch_ms": round(ms, 4), "windows_per_pair": n_win,
        "kernel": kernel + " + ensemble_merge_kernel", "finish_ms": round(fin_ms, 2), "finite_vectors": round(float(np.isfinite(u).mean()), 4),
        "final_fit_rescue": {k: st[k] for k in ("flagged", "rescued", "float32_kept")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes_per_pair": b_alg_pair, "traffic": tr["bytes"] if tr else None,
                     **({"traffic_source": f"profiles/{tr['source']}"} if tr else {}),
                     "secondary": {"bound": "fp32-valu", "achieved_tflops": round(flop_per_pair(ws, n_win) * pairs / (ms * 1e-3) / 1e12, 2),
                                   "peak_tflops": FP32_VALU_TFLOPS}},
    }


def host_fed_rates(lib, sample_u8: np.ndarray, ws, ov) -> dict:
    """PCIe-inclusive rates of lspiv_piv_pairs (frames in pageable host memory -> results in host memory) for the three
    frame dtypes pyorc hands over; never `value`."""
    from pyorc_amd import piv

    out = {}
    for key, arr in (("u8", sample_u8), ("f32", sample_u8.astype(np.float32)), ("f64", sample_u8.astype(np.float64))):
        piv.piv_pairs(arr, ws, ov)  # grows the library's workspaces and pinned ring, touches the pages
        t0 = time.perf_counter()
        piv.piv_pairs(arr, ws, ov)
        out[key] = round((arr.shape[0] - 1) / (time.perf_counter() - t0), 1)
    return out


class _LazyOrthoStack:
    """A stand-in for the dask-backed DataArray pyorc hands to get_ffpiv: slicing along time is free, ``load()`` does per frame what
    ``project_numpy`` does inside a real ``.load()`` (pyorc/project.py:123-161: a nearest-neighbour gather of the camera frame into the
    ortho grid, float64 out) -- plain numpy on one core, GIL released inside the gather and the conversion."""

    def __init__(self, cam: np.ndarray, idx: np.ndarray, lo: int = 0, hi=None):
        self.cam, self.idx, self.lo, self.hi = cam, idx, lo, cam.shape[0] if hi is None else hi
        self.dtype, self.shape = np.dtype(np.float64), (self.hi - self.lo,) + cam.shape[1:]

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b, _ = key.indices(len(self))
            return _LazyOrthoStack(self.cam, self.idx, self.lo + a, self.lo + b)
        return self.load_frame(self.lo + key)

    def load_frame(self, f):
        return np.take(self.cam[f].ravel(), self.idx).astype(np.float64).reshape(self.cam.shape[1:])

    def load(self):
        out = np.empty((len(self),) + self.cam.shape[1:], np.float64)
        for k in range(len(self)):
            out[k].reshape(-1)[:] = np.take(self.cam[self.lo + k].ravel(), self.idx)     # gather, then uint8 -> float64
        return out


def lazy_host_chunk_rates(cam: np.ndarray, ws, ov) -> dict:
    """get_piv over a LAZY host stack with DEFAULT arguments (no ``chunksize=``): the planner cuts loads against host
    memory and the overlap granule, the stack stays resident in HBM, launches go on the anchors (pyorc_amd/resident.py), and the chunk
    executor's depth adapts to the run (pyorc_amd/executor.py).  Against the reference's serial order (``prefetch=0``: load a piece, then
    upload + launch it) and a fixed depth of one.  PCIe- and host-inclusive, never `value`."""
    from pyorc_amd import executor, frames as F

    T, H, W = cam.shape
    idx = np.arange(H * W, dtype=np.int64).reshape(H, W)
    idx = np.roll(idx, 7, axis=1).ravel()                   # a nearest-neighbour plan (a shifted identity: every cell one camera pixel)
    t = np.arange(T) / 30.0
    lazy = _LazyOrthoStack(cam, idx)
    F.get_piv(lazy[:27], ws[0], overlap=ov, time=t[:27], resolution=0.01, prefetch=0)   # workspaces, pinned ring
    out = {}
    ref = None
    for key, depth in (("serial_like_the_reference", 0), ("default_arguments", None), ("prefetch_depth1", 1)):
        t0 = time.perf_counter()
        ds = F.get_piv(lazy, ws[0], overlap=ov, time=t, resolution=0.01, prefetch=depth)
        dt = time.perf_counter() - t0
        st = dict(executor.LAST_STATS)
        same = True if ref is None else all(np.array_equal(ds[k], ref[k], equal_nan=True) for k in ("v_x", "v_y", "corr", "s2n"))
        ref = ds if ref is None else ref
        out[key] = {"pairs_per_s": round((T - 1) / dt, 1), "wall_s": round(dt, 3), "load_s_total": st.get("load_s"),
                    "waited_for_loads_s": st.get("waited_s"), "upload_s": st.get("upload_s"), "launch_s": st.get("launch_s"),
                    "chunks": st.get("chunks"), "depth_per_chunk": st.get("depth_per_chunk"), "same_bits_as_serial": bool(same)}
    out["speedup_default_over_serial"] = round(out["serial_like_the_reference"]["wall_s"] / max(out["default_arguments"]["wall_s"], 1e-9), 2)
    out["plan"] = {k: v for k, v in (st.get("plan") or {}).items() if k in ("load_frames", "align", "max_depth", "source", "peak_host_bytes")}
    out["note"] = (f"{T - 1} pairs of {H}x{W} in {st.get('chunks')} loads planned by get_ffpiv itself: every load is materialised by a per-frame numpy gather + "
                   "float64 conversion (what project_numpy does inside dask's .load()), uploaded to its place in the HBM-resident stack (narrowed to "
                   "float32 while staged) and launched on the kernels' anchors; wall = loads + uploads for the reference's serial loop, "
                   "~ max(loads, uploads) with the loads running ahead")
    return out


def dropin_rates(cam: np.ndarray, maps, dst, ws, ov, t, plan) -> dict:
    """The reference-shaped DROP-IN flow: ``frames.project(method="hip")`` -> ``frames.get_piv(engine="hip")`` on a
    lazy camera stack in 20-frame blocks, i.e. ``pyorc_amd.plugin.project_hip`` (found by name by pyorc's ``Frames.project``, which adds
    its ``fillna(0.0)``) and ``pyorc_amd.velocimetry.get_ffpiv`` with their default arguments.  xarray and dask are absent on the GPU box:
    the lazy DataArray is the double of tests/lazy_doubles.py (blocks computed on a thread pool, dask-style graph names).  Camera frames
    are the normalised uint8 stack already in host memory (decode and ``normalize`` are the caller's dask graph, not this engine's);
    timed: building the projection graph + get_piv, results in host memory.  Compared bit for bit with the reference's data flow on the
    same frames (host stacks, float64 ortho frames into get_piv), and timed next to the generic path (one more layer between project
    and get_piv: the projected blocks come back to the host and go up again)."""
    import sys as _sys

    from pyorc_amd import executor, filters, frames as F, plugin
    from tests import lazy_doubles

    norm = filters.normalize(cam, 15)
    had = _sys.modules.get("xarray")
    _sys.modules["xarray"] = lazy_doubles
    try:
        def run(extra_layer):
            video = lazy_doubles.from_frames(norm, block=20, coords={"time": t})
            ortho = lazy_doubles.frames_project(video, maps, dst, plugin.project_hip)
            if extra_layer:
                ortho = ortho.map_time(lambda blk: blk, "astype")
            return F.get_piv(ortho, ws[0], overlap=ov, time=t, resolution=0.01)

        out, results = {}, {}
        for key, extra in (("dropin_project_hip_get_piv_hip", False), ("dropin_generic_path", True)):
            run(extra)
            t0 = time.perf_counter()
            results[key] = run(extra)
            dt = time.perf_counter() - t0
            st = dict(executor.LAST_STATS)
            out[key] = round((len(cam) - 1) / dt, 1)
            out[key + "_detail"] = {"wall_s": round(dt, 4), "source_loaded": (st.get("plan") or {}).get("source"), "loads": st.get("chunks"),
                                    "load_s": st.get("load_s"), "upload_and_project_s": st.get("upload_s"), "launch_s": st.get("launch_s"),
                                    "waited_for_loads_s": st.get("waited_s")}
    finally:
        if had is None:
            _sys.modules.pop("xarray", None)
        else:
            _sys.modules["xarray"] = had
        plugin.uninstall()
    ref = F.get_piv(plan.project_frames(norm).astype(np.float64), ws[0], overlap=ov, time=t, resolution=0.01)    # host_stacks_float64's data flow
    out["dropin_bit_equal_to_host_stacks_float64"] = bool(all(np.array_equal(results[k][v], ref[v], equal_nan=True)
                                                           for k in results for v in ("v_x", "v_y", "corr", "s2n")))
    return out


def camera_to_velocity_rates(cam: np.ndarray, ws, ov) -> dict:
    """End to end from raw uint8 camera frames in host memory to velocities in host memory, through the accessor-shaped
    entry points: (a) the stages hand HBM-resident stacks to each other (DeviceFrames: one H2D of the camera bytes,
    filters.normalize -> Projection.project_frames -> frames.get_piv, results D2H); (a') the same stages as the fixed chain
    pipeline.CameraToVelocity, in one piece and with the upload streamed in time chunks under the kernels of the previous chunk; (b) every stage returns a host stack
    and the ortho frames reach get_piv as float64, which is what pyorc's project_numpy really hands over."""
    from pyorc_amd import DeviceFrames, filters, frames as F
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    T, H, W = cam.shape
    # the ortho grid at 3/4 of the camera's resolution: 22 % of its cells are then group means of 2+ camera pixels (the same shapes
    # at 1 : 1 have none -- what rounds 2 and 3 timed under this name was a nearest-neighbour-only plan with float32 output)
    Ho, Wo = 3 * H // 4, 3 * W // 4
    maps = projection_maps((H, W), (Ho, Wo), tilt=0.1, seed=1)
    p = Projection((H, W), (Ho, Wo), *maps)
    averaged_cells = int(len(maps[3]))
    t = np.arange(T) / 30.0

    def device_chain():
        d = p.project_frames(filters.normalize(DeviceFrames.from_host(cam), 15))
        return F.get_piv(d, ws[0], overlap=ov, time=t, resolution=0.01)

    def host_chain():
        o = p.project_frames(filters.normalize(cam, 15)).astype(np.float64)
        return F.get_piv(o, ws[0], overlap=ov, time=t, resolution=0.01)

    from pyorc_amd.pipeline import CameraToVelocity
    chain = CameraToVelocity((H, W), (Ho, Wo), *maps, window_size=ws, overlap=ov, normalize_samples=15)
    dropin = dropin_rates(cam, maps, (Ho, Wo), ws, ov, t, p)

    out = {}
    for key, fn in (("device_resident_stages", device_chain), ("streamed_chain", lambda: chain.run(cam, streamed=True)),
                    ("one_piece_chain", lambda: chain.run(cam, streamed=False)), ("host_stacks_float64", host_chain)):
        fn()
        t0 = time.perf_counter()
        r = fn()
        out[key] = round((T - 1) / (time.perf_counter() - t0), 1)
    chain.close()
    # what the two kinds of plan cost once the (normalised) camera stack is in HBM: project + get_piv, results left in HBM.
    # Group means make float32 ortho frames (float32 PIV kernels); a nearest-neighbour-only plan (reducer other than "mean")
    # keeps uint8 frames uint8 (Projection.project_frames, keep_uint8) and the PIV runs its uint8 kernels
    from pyorc_amd import _lib, window
    lib = _lib.load()
    d_norm = filters.normalize(DeviceFrames.from_host(cam), 15)
    pn = Projection((H, W), (Ho, Wo), maps[0], maps[1])
    nr, nc = window.get_array_shape((Ho, Wo), ws, ov)
    d_res = DeviceFrames.empty((4 * (T - 1), nr, nc), np.float32)
    res = {}
    for key, plan, u8 in (("group_means_float32", p, False), ("nearest_only_uint8", pn, True)):
        d_ortho = DeviceFrames.empty((T, Ho, Wo), np.uint8 if u8 else np.float32)

        def fn():
            plan.project_frames_dev(d_norm.ptr, np.uint8, T, d_ortho.ptr, keep_uint8=u8)
            _lib.check(lib.lspiv_piv_pairs_dev(d_ortho.c_ptr, 0 if u8 else 1, T, Ho, Wo, ws[0], ws[1], ov[0], ov[1], -1.0,
                                               d_res.c_ptr, None, None))
        fn()
        _lib.check(lib.lspiv_synchronize())
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        _lib.check(lib.lspiv_synchronize())
        res[key] = round(3 * (T - 1) / (time.perf_counter() - t0), 1)
        del d_ortho
    del d_res
    out["hbm_resident_project_then_piv"] = res
    out["ngwerere_recipe"] = recipe_rates(cam, maps, (Ho, Wo), ws, ov)
    out.update(dropin)
    pn.close()
    del d_norm
    p.close()
    out["note"] = (f"{T - 1} pairs of {H}x{W} uint8 camera frames in pageable host memory -> normalize(15) -> orthoprojection "
                   f"to a {Ho}x{Wo} grid (synthetic homography; {averaged_cells} of its {Ho * Wo} cells are group means of 2+ camera pixels, the rest "
                   f"nearest neighbour) -> get_piv {ws[0]}x{ws[1]}; PCIe-inclusive, never `value`")
    return out


def recipe_rates(cam: np.ndarray, maps, ortho_shape, ws, ov) -> dict:
    """The `frames:` section of the reference's own recipe in its own order (examples/ngwerere/ngwerere.yml:5-11; the service appends
    `project` last, pyorc/service/velocimetry.py:537-538): normalize -> edge_detect(1, 2) -> minmax(-5, 5) -> project -> get_piv.  What
    project sees is then a FLOAT32 camera stack (the float32 tiles of round 6).  (a) the fixed chain from uint8 camera frames in host
    memory to velocities in host memory, upload streamed under the kernels; (b) the stages once the camera stack is in HBM, each timed
    by HIP events, results left in HBM (minmax rides in edge_detect's store: lspiv_edge_detect_clip_dev; as a pass of its own it
    cost 0.73 ms of the chain's 3.4)."""
    import ctypes as C

    from pyorc_amd import DeviceFrames, _lib, filters, window
    from pyorc_amd.pipeline import CameraToVelocity
    from pyorc_amd.project import Projection

    T, H, W = cam.shape
    Ho, Wo = ortho_shape
    lib = _lib.load()
    out = {}
    with CameraToVelocity((H, W), (Ho, Wo), *maps, window_size=ws, overlap=ov, normalize_samples=15, edge_detect=(1, 2), minmax=(-5, 5)) as chain:
        chain.run(cam)
        t0 = time.perf_counter()
        chain.run(cam)
        out["host_to_host_pairs_per_s"] = round((T - 1) / (time.perf_counter() - t0), 1)
    p = Projection((H, W), (Ho, Wo), *maps)
    nr, nc = window.get_array_shape((Ho, Wo), ws, ov)
    d_cam = DeviceFrames.from_host(cam)
    d_edge = DeviceFrames.empty((T, H, W), np.float32)
    d_ortho = DeviceFrames.empty((T, Ho, Wo), np.float32)
    d_res = DeviceFrames.empty((4 * (T - 1), nr, nc), np.float32)
    d_n = DeviceFrames.empty((T, H, W), np.uint8)
    stages = (("normalize", lambda: _lib.check(lib.lspiv_normalize_dev(d_cam.c_ptr, T, H, W, 15, d_n.c_ptr, None))),
              ("edge_detect_minmax", lambda: _lib.check(lib.lspiv_edge_detect_clip_dev(d_n.c_ptr, 0, T, H, W, 3, 5, -5.0, 5.0, d_edge.c_ptr, None))),
              ("project_float32", lambda: p.project_frames_dev(d_edge.ptr, np.float32, T, d_ortho.ptr)),
              ("get_piv", lambda: _lib.check(lib.lspiv_piv_pairs_dev(d_ortho.c_ptr, 1, T, Ho, Wo, ws[0], ws[1], ov[0], ov[1], -1.0, d_res.c_ptr, None, None))))
    ev = [C.c_void_p() for _ in range(len(stages) + 1)]
    for e in ev:
        _lib.check(lib.lspiv_event_create(C.byref(e)))
    for _ in range(2):                                  # the second pass is the one read
        _lib.check(lib.lspiv_synchronize())
        _lib.check(lib.lspiv_event_record(ev[0]))
        for i, (_, fn) in enumerate(stages):
            fn()
            _lib.check(lib.lspiv_event_record(ev[i + 1]))
        _lib.check(lib.lspiv_synchronize())
    ms = {}
    for i, (name, _) in enumerate(stages):
        v = C.c_float()
        _lib.check(lib.lspiv_event_elapsed_ms(ev[i], ev[i + 1], C.byref(v)))
        ms[name] = round(v.value, 3)
    for e in ev:
        _lib.check(lib.lspiv_event_destroy(e))
    p.close()
    out["hbm_resident_stage_ms"] = ms
    out["hbm_resident_pairs_per_s"] = round((T - 1) / (sum(ms.values()) * 1e-3), 1)
    out["note"] = (f"{T - 1} pairs; the reference's recipe order: normalize(15) -> edge_detect(1, 2) -> minmax(-5, 5) -> project (float32 in, "
                   f"{Ho}x{Wo} float32 out) -> get_piv {ws[0]}x{ws[1]}; PCIe-inclusive / host-inclusive, never `value`")
    return out


def c_oracle_uv(sample, ws, ov, cores):
    """u, v of the C port for the first pairs of the sample (what cpu_baseline.numpy_pocketfft states its agreement with)."""
    from oracle import c_oracle

    n = min(sample.shape[0] - 1, 2 * cores)
    u, v, _, _ = c_oracle.piv_pairs(sample[:n + 1], ws, ov, nthreads=cores)
    return u, v


def rows_block(lib) -> list:
    """The HBM-bound rows around the path, each with its own roofline block: tools/rows_launch.py."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("rows_launch", os.path.join(ROOT, "tools", "rows_launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.measure_rows(lib)


def open_comm(rank: int, world: int):
    """The communicator of the run: RCCL unless LSPIV_COMM says otherwise.  RCCL has never met two ranks on the boxes this was built on, so
    a node on which it cannot be brought up (no librccl, an IPC mode the driver refuses, ...) must still yield a line that SAYS so: every
    rank tries ncclCommInitRank plus one tiny device all-gather, the ranks then vote through files next to the id file (the rendezvous
    they already share), and unless all of them succeeded all of them fall back to the shared-memory transport.  Returns (comm, None) or
    (comm over shm, the first error message)."""
    import ctypes as C

    from pyorc_amd import comm as cm

    want = os.environ.get("LSPIV_COMM", "rccl")
    if want != "rccl" or world == 1:
        return cm.Comm(rank, world), None
    base = cm.default_id_file()
    nonce = cm._job_nonce().hex()
    c, err = None, None
    try:
        c = cm.Comm(rank, world, "rccl", id_file=base, timeout=180.0)
        lib = _lib.load()
        d = C.c_void_p()
        _lib.check(lib.lspiv_dev_malloc(C.byref(d), 4 * (world + 1)))
        try:
            c.allgather_dev(d.value, d.value + 4, 1, np.float32)      # what fails here would fail in the first step
            _lib.check(lib.lspiv_synchronize())
        finally:
            lib.lspiv_dev_free(d)
    except Exception as e:                                            # noqa: BLE001 -- any failure is a vote
        err = f"rank {rank}: {type(e).__name__}: {e}"
    mine = f"{base}.vote.{rank}"
    with open(mine, "w") as fh:
        fh.write(f"{nonce}\n{err or 'ok'}\n")
    votes, t0 = {}, time.time()
    while len(votes) < world:
        for r in range(world):
            if r in votes:
                continue
            try:
                txt = open(f"{base}.vote.{r}").read().split("\n")
                if len(txt) >= 2 and txt[0] == nonce:
                    votes[r] = txt[1]
            except OSError:
                pass
        if len(votes) < world:
            if time.time() - t0 > 240.0:
                raise TimeoutError(f"rank {rank}: only {sorted(votes)} of {world} ranks reported on their communicator" + (f"; own error: {err}" if err else ""))
            time.sleep(0.02)
    bad = [v for _, v in sorted(votes.items()) if v != "ok"]
    if not bad:
        c.barrier()                                                   # every rank has read every vote
        try:
            os.unlink(mine)
        except OSError:
            pass
        return c, None
    if c is not None:
        c.close()
    print(f"[bench] rank {rank}: RCCL is not usable on every rank ({bad[0]}); falling back to the shared-memory transport", file=sys.stderr, flush=True)
    c = cm.Comm(rank, world, "shm", id_file=base + ".shm")
    c.barrier()
    try:
        os.unlink(mine)
    except OSError:
        pass
    return c, bad[0]


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and wait for them."""
    tmp = tempfile.mkdtemp(prefix="lspiv_bench_")
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   LSPIV_COMM_ID_FILE=os.path.join(tmp, "comm_id"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        deadline = time.time() + float(os.environ.get("LSPIV_BENCH_TIMEOUT_S", "1800"))
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    rc = rc or code
            if rc or time.time() > deadline:
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:   # exact PIDs of the ranks started above
            if p.poll() is None:
                p.kill()
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    return rc


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))
    # The JSON line must be the only thing on stdout, and native libraries write there too (RCCL prints a version banner
    # through C stdio, flushed at exit, i.e. AFTER anything Python printed): keep the real stdout for the JSON line and
    # point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
    same_device = bool(os.environ.get("LSPIV_BENCH_SAME_DEVICE"))  # plumbing test of N > 1 on a 1-GPU box (not a measurement)
    if same_device:
        local_rank = 0
        os.environ.setdefault("LSPIV_COMM", "shm")   # RCCL refuses two ranks on one GPU
    lib = _lib.load()
    _lib.require_device()
    _lib.check(lib.lspiv_set_device(local_rank))
    comm = None
    rccl_error = None
    use_comm = world > 1 or bool(os.environ.get("LSPIV_BENCH_FORCE_COMM"))  # FORCE_COMM: 1-rank RCCL plumbing test
    if use_comm:
        comm, rccl_error = open_comm(rank, world)

    H, W = a.height, a.width
    ws, ov = (a.window, a.window), (a.overlap, a.overlap)
    n_rows, n_cols = window.get_array_shape((H, W), ws, ov)
    n_win = n_rows * n_cols

    # ---- which pairs this rank owns ----------------------------------------------------------
    # N = 1 without a communicator: one launch over --pairs pairs.  With a communicator the step IS the library's sharded path
    # (pyorc_amd.shard.ShardedPivDev: this rank's block of the time axis resident in HBM, kernels anchored at their absolute pair
    # index, the packed result block all-gathered one step behind the kernels) -- weak: every rank owns --pairs pairs of a
    # world x --pairs stack (blocks cut at multiples of --pairs); --strong: --strong-pairs pairs in total, cut on the walking
    # kernels' anchors (shard.pair_block).
    plan = None
    if use_comm:
        from pyorc_amd import shard

        total_pairs = a.strong_pairs if a.strong else a.pairs * world
        plan = shard.ShardedPivDev(comm, total_pairs, (H, W), ws, ov, align=None if a.strong else a.pairs, record_timings=True)
        my_pairs, pair_offset = plan.p_local, plan.a
    else:
        total_pairs = my_pairs = a.pairs
        pair_offset = 0
    T = my_pairs + 1
    n_tiles = my_pairs * n_win

    # ---- device-resident synthetic stack + result block ---------------------------------------
    def dev_alloc(nbytes):
        p = C.c_void_p()
        _lib.check(lib.lspiv_dev_malloc(C.byref(p), max(int(nbytes), 256)))
        return p

    from pyorc_amd.device import DeviceFrames

    frames = DeviceFrames.empty((T, H, W), np.uint8)
    d_frames = frames.c_ptr
    d_out = dev_alloc(4 * n_tiles * 4)
    _lib.check(lib.lspiv_synth_particles_dev(d_frames, T, H, W, a.seed + rank, 0.02))
    _lib.check(lib.lspiv_synchronize())  # the generator ran on the library's stream; the steps use other ones

    def launch_all(out=None, stream=None):
        if my_pairs:
            _lib.check(lib.lspiv_piv_pairs_dev_at(d_frames, 0, T, H, W, ws[0], ws[1], ov[0], ov[1], -1.0, pair_offset, out or d_out, None, stream))

    if plan is None:
        step = launch_all

        def drain():
            pass
    else:
        def step():
            plan.step(frames)

        drain = plan.drain

    def sync():
        _lib.check(lib.lspiv_synchronize())

    def barrier():
        if comm is not None:
            comm.barrier()

    for _ in range(a.warmup):
        step()
    drain(); sync(); barrier(); sync()
    if plan is not None:
        plan.timings()          # drop the warm-up steps' marks
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain(); sync(); barrier(); sync()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = float(comm.allreduce(np.array([dt], dtype=np.float64), 1)[0])   # MAX over ranks
    step_marks = plan.timings() if plan is not None else None

    # ---- live kernel timing with HIP events on the launch stream (roofline leg, N-independent) --
    reps = max(3, min(a.steps, 10))
    launch_ms = time_launches(lib, launch_all, reps)          # PIV kernel + the two rescue kernels: what a step issues
    # the dominant kernel alone (what roofline.achieved and the rocprofv3 per-kernel average are about): the same launch with
    # the rescue pass switched off -- the kernel still evaluates its flags, it just appends nothing and no rescue kernel follows
    _lib.set_option("rescue", 0)
    try:
        kernel_ms_rescue_off = time_launches(lib, launch_all, reps)
    finally:
        _lib.set_option("rescue", 1)
    # ... and the dominant kernel INSIDE the launch as issued (rescue on: it appends its records), by the library's events right
    # around it: what rocprofv3 --kernel-trace reports for it, and what roofline.achieved is computed from (the
    # rescue-off figure was ~3 % kinder than the profile the line cites)
    kernel_ms, kernel_ms_each = time_kernel_only(lib, launch_all, reps)
    if kernel_ms is None:
        kernel_ms = launch_ms
    launch_all()                                              # leave d_out as a full launch (with the rescue pass) produces it
    sustained = None
    if world == 1 and a.sustained_s > 0:
        sustained = sustained_run(lib, step, sync, total_pairs, a.sustained_s)
        launch_all()

    # float64 rescue pass: how many windows of one launch the kernels flagged (the launches above ran on the library's stream)
    rescue = None
    if hasattr(lib, "lspiv_rescue_stats"):
        st = (C.c_int64 * 5)()
        _lib.check(lib.lspiv_rescue_stats(None, st))
        rescue = {"enabled": bool(_lib.get_option("rescue")), "fit_windows_per_launch": int(st[0]), "amb_windows_per_launch": int(st[1]),
                  "share_of_windows": round((st[0] + st[1]) / max(n_tiles, 1), 6),
                  "note": "windows whose float32 peak fit the kernel flags as ill-conditioned are re-evaluated from the frames in "
                          "float64 (csrc/piv_rescue.hip); the two rescue kernels run inside the timed region and inside kernel_ms"}
    gather_ms = None
    if plan is not None and plan.k > 0:
        # one all-gather of a result block on its own, by events on the gather stream (outside the timed region)
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(lib.lspiv_event_create(C.byref(e0)))
        _lib.check(lib.lspiv_event_create(C.byref(e1)))
        sync()
        _lib.check(lib.lspiv_event_record_on(e0, plan.comm_s))
        comm.allgather_dev(plan.send[0].ptr, plan.recv[0].ptr, plan.count, np.float32, plan.comm_s.value)
        _lib.check(lib.lspiv_event_record_on(e1, plan.comm_s))
        _lib.check(lib.lspiv_stream_synchronize(plan.comm_s))
        ms = C.c_float()
        _lib.check(lib.lspiv_event_elapsed_ms(e0, e1, C.byref(ms)))
        gather_ms = round(ms.value, 4)
        _lib.check(lib.lspiv_event_destroy(e0))
        _lib.check(lib.lspiv_event_destroy(e1))

    # ---- one STRONG pass next to the weak one (north_star says "strong scaling"; the driver runs the default) ----
    strong = None
    if plan is not None and not a.strong and a.strong_pairs > 0:
        from pyorc_amd import shard

        plan_s = shard.ShardedPivDev(comm, a.strong_pairs, (H, W), ws, ov, align=None)
        need = plan_s.p_local + 1 if plan_s.p_local else 0
        if need <= T:
            block_s = frames[:need]
        else:                                   # fewer ranks than the total was meant for: this rank's block is longer than the weak one
            block_s = DeviceFrames.empty((need, H, W), np.uint8)
            _lib.check(lib.lspiv_synth_particles_dev(block_s.c_ptr, need, H, W, a.seed + rank, 0.02))
            _lib.check(lib.lspiv_synchronize())
        for _ in range(2):
            plan_s.step(block_s)
        plan_s.drain(); sync(); barrier(); sync()
        t0s = time.perf_counter()
        for _ in range(a.steps):
            plan_s.step(block_s)
        plan_s.drain(); sync(); barrier(); sync()
        dts = time.perf_counter() - t0s
        dts = float(comm.allreduce(np.array([dts], dtype=np.float64), 1)[0])   # MAX over ranks
        strong = {"strong_pairs_total": a.strong_pairs, "strong_pairs_rank0": plan_s.p_local, "strong_ms_per_step": round(dts / a.steps * 1e3, 4),
                  "strong_pairs_per_s": round(a.strong_pairs * a.steps / dts, 2),
                  "strong_note": "the same step loop with --strong-pairs pairs IN TOTAL cut over the ranks on the kernels' anchors (what `--strong` times as "
                                 "`value`), run after the weak timed region: north_star's strong-scaling reading beside the default weak one"}
        plan_s.close()
        del block_s

    dist_check = None
    if plan is not None and plan.k > 0:
        # this rank's slice of the all-gathered block must equal its own single-launch result bit for bit; cheap, outside the
        # timed region (launch_all above wrote the same pairs, at the same absolute pair index, into d_out)
        plan.step(frames)
        whole = np.empty((4, my_pairs, n_rows, n_cols), dtype=np.float32)
        if my_pairs:
            _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(whole), d_out, whole.nbytes))
        mine = plan.gathered_host()[:, plan.a:plan.b]
        ok = np.array([1.0 if np.array_equal(mine.view(np.uint32), whole.view(np.uint32)) else 0.0], dtype=np.float32)
        dist_check = bool(comm.allreduce(ok, 0)[0] == world)   # true on every rank
    if rank != 0:
        if comm is not None:
            comm.barrier()
            comm.close()
        return

    pairs_per_s = total_pairs * a.steps / dt
    is_c2 = (a.window, a.overlap, H, W) == (32, 16, 1080, 1920)
    out = {
        # BASELINE.json's metric; a non-default --window / --overlap / --height / --width run says what it measured
        "metric": ("PIV frame-pairs/sec, 1080p 32x32@50% overlap (Mvectors/sec in config)" if is_c2 else
                   f"PIV frame-pairs/sec, {H}x{W} frames, {a.window}x{a.window} windows @ overlap {a.overlap} (not the BASELINE.json metric)"),
        "value": round(pairs_per_s, 2),
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if (a.strong and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {H}x{W} uint8 particle stack, "
                        + (f"{total_pairs} frame-pairs in total cut over {world} GPUs, " if (a.strong and world > 1) else f"{a.pairs} frame-pairs per GPU, ") +
                        f"{a.window}x{a.window} windows @ overlap {a.overlap} ("
                        + ("BASELINE.json configs[1]" if is_c2 and a.pairs == 1000 else "a variation of BASELINE.json configs[1]")
                        + ("; configs[4] sharding" if world > 1 else "") + ")",
            "frame_dtype": "u8",
            "windows_per_pair": n_win,
            "mvectors_per_s": round(pairs_per_s * n_win / 1e6, 3),
            "parallelism": (f"time-block shard x{world} (pyorc_amd.shard.ShardedPivDev), {comm.transport.upper()} all-gather of the "
                            f"(4,t,y,x) result block through the C ABI (no torch)") if comm is not None else "single GPU",
            "scaling_denominator": ("--strong: the total is fixed at --strong-pairs; speed-up at N GPUs = value(N) / value(1) of `bench.py "
                                    "--strong --gpus N`" if a.strong else
                                    "per-GPU work fixed at --pairs: speed-up at N GPUs = value(N) / value(1) of this command "
                                    "(north_star's >= 6.5x at 8 GPUs = 8000 pairs on 8 GPUs vs 1000 pairs on 1); `--strong` cuts a "
                                    "fixed 8000-pair total over the ranks instead"),
        },
        "roofline": roofline_block(lib, kernel_ms, a.pairs, H, W, a.window, a.overlap, n_win, launch_ms),
    }
    out["roofline"]["kernel_timing"] = ("mean of HIP-event pairs recorded by the library right around the dominant kernel in " + str(len(kernel_ms_each)) +
                                        " launches issued as a caller issues them (rescue pass on), on the launch stream -- the per-kernel "
                                        "duration of `rocprofv3 --kernel-trace`; profiles/*_summary.json hold the trace's own mean")
    out["roofline"]["kernel_ms_same_launch_rescue_off"] = round(kernel_ms_rescue_off, 4)
    if sustained is not None:
        # value / ms_per_step time a burst of --steps launches (0.12 s); this is the same step loop held for >= 10 s
        out["sustained_pairs_per_s"] = sustained["pairs_per_s"]
        out["config"]["sustained"] = {**sustained, "note": "the timed step loop repeated for >= --sustained-s seconds after the timed region; "
                                      "clock and socket power by rocm-smi once a second (first second dropped); `value` is the burst"}
    out["config"]["binary"] = _lib.binary_provenance(lib)   # hashes compiled into the loaded .so next to the tree's
    if rescue is not None:
        out["config"]["rescue"] = rescue
    if comm is not None:
        km = step_marks["kernel_ms"][-a.steps:] if step_marks else []
        gm = step_marks["gather_ms"][-a.steps:] if step_marks else []
        lag = step_marks["gather_end_after_kernel_end_ms"][-a.steps:] if step_marks else []
        mean = lambda x: round(float(np.mean(x)), 4) if len(x) else None   # noqa: E731
        out["config"]["comm"] = {
            "transport": comm.transport, "ranks_reported_by_transport": comm.backend_ranks,
            **({"rccl_error": rccl_error, "note": "RCCL could not be brought up on every rank: this line ran over the shared-memory transport "
                                                  "(results cross the host) -- a diagnosis, not the xGMI figure"} if rccl_error else {}),
            "path": "pyorc_amd.shard.ShardedPivDev (the library's device-resident sharded path; bench.py only times it)",
            "mode": "strong" if a.strong else "weak", "pairs_total": total_pairs, "pairs_rank0": my_pairs,
            "allgather_matches_single_launch": dist_check,
            "allgather_bytes_per_rank_per_step": 4 * plan.p_max * n_win * 4,
            "allgather_bytes_received_per_rank_per_step": (world - 1) * 4 * plan.p_max * n_win * 4,
            "survey_8e_bytes_per_rank": "SURVEY.md section 8e: 1000 pairs x 7854 vectors x 16 B = 125.7 MB per rank and step at the C5 shape",
            "allgather_ms_alone": gather_ms,
            # rank 0, by HIP events per timed step: this rank's kernels (PIV + rescue) while the previous step's gather is in
            # flight on the other stream, that gather itself, and what of the step is not covered by the kernels
            "kernel_ms_while_gather_in_flight": mean(km), "gather_ms_overlapped": mean(gm),
            "gather_end_after_kernel_end_ms": mean(lag),
            "exposed_comm_ms": round(dt / a.steps * 1e3 - float(np.mean(km)), 4) if len(km) else None,
            "kernel_ms_alone": round(launch_ms, 4),
            "gather_stream_priority": "high" if plan.gather_priority > 0 else ("low" if plan.gather_priority < 0 else "default"),
            **(strong or {}),
            # NCCL_MAX_NCHANNELS: what was in force when the RCCL communicator was created (pyorc_amd.comm sets its default only
            # around ncclCommInitRank and restores the environment afterwards)
            "rccl_env": {**{k: os.environ.get(k) for k in ("NCCL_MIN_NCHANNELS", "LSPIV_RCCL_MAX_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO")
                            if os.environ.get(k) is not None},
                         **({"NCCL_MAX_NCHANNELS": _comm_mod._rccl_channel_cap.applied} if _comm_mod._rccl_channel_cap.applied else {})},
            **({"same_device_plumbing_test": True} if same_device else {})}
    # ---- CPU baseline on a bounded sample of the same stack (rank 0, N = 1 only) --------------
    if world == 1 and a.cpu_pairs != 0:
        from oracle import cpu_baseline as cb   # test infrastructure: baseline / parity leg only

        n_s = cb.default_sample_pairs(H, W, ws) if a.cpu_pairs < 0 else a.cpu_pairs
        n_s = min(n_s, a.pairs)
        sample = np.empty((n_s + 1, H, W), dtype=np.uint8)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(sample), d_frames, sample.nbytes))
        gpu_block = np.empty((4, a.pairs, n_rows, n_cols), dtype=np.float32)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(gpu_block), d_out, gpu_block.nbytes))
        base = cb.run(sample, ws, ov, gpu_block[:, :n_s])
        if is_c2 and not a.no_extras and n_s >= 3:
            # ensemble mode on the full window grids of configs[1] / [2], three pairs: the product against the numpy oracle
            from pyorc_amd import piv

            base["ensemble_parity"] = []
            for ews, eov in (((32, 32), (16, 16)), ((64, 64), (48, 48))):
                ens = piv.Ensemble((H, W), ews, eov)
                ens.accumulate(sample[:4], 0.2, 3.0)
                eu, ev, _ = ens.finish(0.2, 1)
                ens.close()
                base["ensemble_parity"].append(cb.ensemble_parity(sample[:4], ews, eov, eu, ev, 0.2, 3.0, 0.2))
        if not a.no_extras:
            # the numpy / pocketfft reading of the same sample beside the C port (one process per core)
            try:
                ref_uv = c_oracle_uv(sample, ws, ov, base["cores"])
                base["numpy_pocketfft"] = cb.numpy_pocketfft(sample, ws, ov, check=ref_uv)
            except Exception as exc:   # a baseline leg must not take the line down
                base["numpy_pocketfft"] = {"error": f"{type(exc).__name__}: {exc}"}
        out["cpu_baseline"] = base
        if base.get("value"):
            out["config"]["speedup_vs_cpu_baseline"] = round(pairs_per_s / base["value"], 1)
    # ---- the other single-GPU BASELINE configs + host-fed rates (N = 1, default shape only) ----
    if world == 1 and is_c2 and not a.no_extras:
        others = [other_config(lib, "BASELINE.json configs[2]: 1080p, 64x64 windows @ 75 % overlap, same stack",
                               d_frames, a.pairs, H, W, 64, 48, parity_pairs=a.parity_pairs if a.cpu_pairs != 0 else 0)]
        ensembles = [ensemble_config(lib, "ensemble correlation (ensemble_corr=True), 1080p, 32x32 @ 50 %, same stack", d_frames, a.pairs, H, W, 32, 16),
                     ensemble_config(lib, "ensemble correlation (ensemble_corr=True), 1080p, 64x64 @ 75 %, same stack", d_frames, a.pairs, H, W, 64, 48)]
        sample = np.empty((min(a.pairs, 200) + 1, H, W), dtype=np.uint8)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(sample), d_frames, sample.nbytes))
        frames = d_frames = None                      # the 1080p stack goes back (DeviceFrames' caching allocator) ...
        from pyorc_amd.device import release_pool

        release_pool()                                # ... and from there to the driver, before the 8.3 GB 4K stack is made
        H4, W4 = 2160, 3840
        d4 = dev_alloc((a.pairs + 1) * H4 * W4)
        _lib.check(lib.lspiv_synth_particles_dev(d4, a.pairs + 1, H4, W4, a.seed + 2, 0.02))
        others.append(other_config(lib, "BASELINE.json configs[3]: 4K (2160x3840), 32x32 windows @ 50 % overlap",
                                   d4, a.pairs, H4, W4, 32, 16, parity_pairs=a.parity_pairs if a.cpu_pairs != 0 else 0))
        _lib.check(lib.lspiv_dev_free(d4))
        out["config"]["other_configs"] = others + ensembles
        out["config"]["host_fed_pairs_per_s"] = {
            **host_fed_rates(lib, sample, ws, ov),
            "note": f"lspiv_piv_pairs on {sample.shape[0] - 1} pairs in pageable host memory, PCIe-inclusive; never `value`"}
        out["config"]["camera_to_velocity_pairs_per_s"] = camera_to_velocity_rates(sample, ws, ov)
        # (a 720p crop: 3 476 windows, anchors every 25 pairs -- 1080p grids are cut on 125 pairs, two chunks of this sample)
        out["config"]["lazy_host_chunks"] = lazy_host_chunk_rates(np.ascontiguousarray(sample[:, :720, :1280]), ws, ov)
        try:
            out["config"]["rows"] = rows_block(lib)
        except Exception as exc:
            out["config"]["rows"] = {"error": f"{type(exc).__name__}: {exc}"}
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
