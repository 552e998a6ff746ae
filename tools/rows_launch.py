"""The HBM-bound rows around the hot path on device-resident 1080p stacks: one launch per call, timed by HIP events on
the library's stream.  bench.py imports ``measure_rows`` for the ``config.rows`` rooflines of its JSON line; as a script it runs ONE row in
a loop so that ``tools/profile.sh`` can take its rocprofv3 passes:

    PROFILE_KF='--kernel-include-regex project_' SUMMARY_KERNELS='project_' PROFILE_CMD="python tools/rows_launch.py project 30" bash tools/profile.sh r06_project

Algorithmic bytes per frame: every input sample read once, every output sample written once -- index maps, weight
tables and second passes over the same frame are NOT credited.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyorc_amd import _lib  # noqa: E402

H, W = 1080, 1920
HO, WO = 810, 1440          # the ortho grid of bench.py's camera -> velocity legs (3/4 of the camera's resolution)
if os.environ.get("ROWS_ORTHO"):   # another ortho grid for the projection rows (A/B sessions): ROWS_ORTHO=540x960
    HO, WO = (int(v) for v in os.environ["ROWS_ORTHO"].split("x"))
HBM_PEAK_GBS = 8000.0


def _alloc(lib, nbytes):
    p = C.c_void_p()
    _lib.check(lib.lspiv_dev_malloc(C.byref(p), max(int(nbytes), 256)))
    return p


def _cv_plan():
    """A mild lens + a perspective warp onto the (HO, WO) grid: what Frames.project(method="cv") builds for a tilted camera."""
    from pyorc_amd.project import ProjectionCV

    K = np.array([[1500.0, 0, W / 2], [0, 1500.0, H / 2], [0, 0, 1]])
    dist = np.array([-0.12, 0.03, 0.001, -0.0005, 0.0])
    M = np.array([[0.78, 0.05, -20.0], [0.01, 0.80, -15.0], [1.5e-5, 4.0e-5, 1.0]])
    return ProjectionCV((H, W), (HO, WO), K, dist, M)


def build_rows(lib, T: int):
    """name -> (launch(), algorithmic bytes per launch, frames per launch, kernel-name regex for the profile summary, description, cleanup)."""
    from pyorc_amd.project import Projection
    from pyorc_amd.synth import projection_maps

    n, no = H * W, HO * WO
    d_cam = _alloc(lib, T * n)
    _lib.check(lib.lspiv_synth_particles_dev(d_cam, T, H, W, 3, 0.02))
    d_f32 = _alloc(lib, T * max(n, no) * 4)
    d_u8 = _alloc(lib, T * n)
    d_camf = _alloc(lib, T * n * 4)                          # float32 camera frames: the edge-detected stack the Ngwerere recipe projects
    _lib.check(lib.lspiv_edge_detect_dev(d_cam, 0, T, H, W, 3, 5, d_camf, None))
    maps = projection_maps((H, W), (HO, WO), tilt=0.1, seed=1)
    p_mean = Projection((H, W), (HO, WO), *maps)
    p_nn = Projection((H, W), (HO, WO), maps[0], maps[1])
    p_cv = _cv_plan()
    rows = {
        "project": (lambda: p_mean.project_frames_dev(d_cam.value, np.uint8, T, d_f32.value), T * (n + 4 * no), T, "project_",
                    f"Frames.project(method='numpy', reducer='mean'): {H}x{W} uint8 camera -> {HO}x{WO} float32 ortho, quad-window plan + group means"),
        "project_nn": (lambda: p_nn.project_frames_dev(d_cam.value, np.uint8, T, d_f32.value), T * (n + 4 * no), T, "project_",
                       "Frames.project(method='numpy') with a reducer other than 'mean': nearest neighbour only, the uint8 quad-window plan, float32 out"),
        "project_u8": (lambda: p_nn.project_frames_dev(d_cam.value, np.uint8, T, d_u8.value, keep_uint8=True), T * (n + no), T, "project_",
                       "nearest-neighbour-only plan (reducer other than 'mean'), uint8 in -> uint8 out"),
        "project_f32": (lambda: p_mean.project_frames_dev(d_camf.value, np.float32, T, d_f32.value), T * (4 * n + 4 * no), T, "project_",
                        f"Frames.project(method='numpy', reducer='mean') of FLOAT32 camera frames (after edge_detect / minmax, the Ngwerere recipe's order): {H}x{W} -> {HO}x{WO}"),
        "edge_detect": (lambda: _lib.check(lib.lspiv_edge_detect_dev(d_cam, 0, T, H, W, 3, 5, d_f32, None)), T * n * 5, T, "blur_|edge_",
                        "Frames.edge_detect(wdw_1=1, wdw_2=2): difference of two Gaussian blurs (3x3, 5x5), uint8 -> float32"),
        "project_cv": (lambda: _lib.check(lib.lspiv_project_cv_frames_dev(p_cv._h, d_cam, 0, T, d_u8, None)), T * (n + no), T, "remap_",
                       "Frames.project(method='cv'): undistort + warpPerspective as two fixed-point bilinear remaps, uint8"),
        "project_cv_f32": (lambda: _lib.check(lib.lspiv_project_cv_frames_dev(p_cv._h, d_camf, 1, T, d_f32, None)), T * (4 * n + 4 * no), T, "remap_",
                           "Frames.project(method='cv') of FLOAT32 frames (after edge_detect): two float32 bilinear remaps"),
        "time_diff": (lambda: _lib.check(lib.lspiv_time_diff_dev(d_cam, 0, T, H, W, 0.0, 0, d_f32, None)), T * n + (T - 1) * n * 4, T - 1, "time_diff_",
                      "Frames.time_diff: uint8 -> float32 difference of consecutive frames (every frame read once, algorithmically)"),
        "normalize": (lambda: _lib.check(lib.lspiv_normalize_dev(d_cam, T, H, W, 15, d_u8, None)), T * n * 2, T, "norm|sample_mean|frame_minmax",
                      "Frames.normalize(samples=15): uint8 -> uint8 (read once + written once credited; the per-frame min / max needs a second look at every frame)"),
        "smooth": (lambda: _lib.check(lib.lspiv_gaussian_blur_dev(d_cam, 0, T, H, W, 5, d_f32, None)), T * n * 5, T, "blur_",
                   "Frames.smooth(wdw=2): cv2.GaussianBlur 5x5, uint8 -> float32"),
        "edge_detect_6_10": (lambda: _lib.check(lib.lspiv_edge_detect_dev(d_cam, 0, T, H, W, 13, 21, d_f32, None)), T * n * 5, T, "blur_|edge_",
                             "Frames.edge_detect(wdw_1=6, wdw_2=10), the reference's user-guide example: Gaussian blurs 13x13 and 21x21 (run-time radii), uint8 -> float32"),
        "smooth_f32": (lambda: _lib.check(lib.lspiv_gaussian_blur_dev(d_camf, 1, T, H, W, 5, d_f32, None)), T * n * 8, T, "blur_",
                       "Frames.smooth(wdw=2) of FLOAT32 frames (after time_diff / a float video): float32 -> float32"),
        "edge_detect_f32": (lambda: _lib.check(lib.lspiv_edge_detect_dev(d_camf, 1, T, H, W, 3, 5, d_f32, None)), T * n * 8, T, "blur_|edge_",
                            "Frames.edge_detect(wdw_1=1, wdw_2=2) of FLOAT32 frames: float32 -> float32"),
    }

    def cleanup():
        for pl in (p_mean, p_nn, p_cv):
            pl.close()
        for p in (d_cam, d_f32, d_u8, d_camf):
            lib.lspiv_dev_free(p)

    return rows, cleanup


def row_read_bytes(name: str, T: int) -> int:
    """Input bytes a launch of this row cannot avoid reading (every input sample once): what FETCH_SIZE is calibrated against."""
    n = H * W
    return T * n * (4 if name in ("project_f32", "smooth_f32", "edge_detect_f32", "project_cv_f32") else 1)     # float32 camera frames


def time_launches(lib, launch, reps: int) -> float:
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


def row_traffic(name: str, T: int):
    """HBM bytes per launch from the committed counter passes of this row (profiles/r06_rows_<name>_summary.json), when they were taken
    on the library sources of this tree (the summary's `source_hash`) and on the same launch shape."""
    import json

    f = os.path.join(ROOT, "profiles", f"r06_rows_{name}_summary.json")
    try:
        d = json.load(open(f))
    except (OSError, ValueError):
        return None
    if d.get("launch", {}).get("frames") != T or d.get("rows_source_hash") != rows_source_hash():
        return None
    tot = sum(k.get("hbm_traffic_bytes", 0.0) * (k.get("launches_per_call", 1.0)) for k in d.get("kernels", {}).values() if "hbm_traffic_bytes" in k)
    return {"bytes": round(tot), "source": os.path.basename(f), "kernels": {k: v.get("trace", {}) for k, v in d.get("kernels", {}).items()}} if tot else None


def rows_source_hash() -> str:
    """sha256 (16 hex digits) over the sources of the row kernels: what a committed r06_rows_* summary is keyed to."""
    import hashlib

    h = hashlib.sha256()
    for f in ("project.hip", "filters.hip"):
        h.update(open(os.path.join(ROOT, "pyorc_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measure_rows(lib, T: int = 201, reps: int = 5, only=None) -> list:
    rows, cleanup = build_rows(lib, T)
    out = []
    try:
        for name, (launch, b_alg, frames, regex, what) in rows.items():
            if only and name not in only:
                continue
            launch(); launch()
            ms = time_launches(lib, launch, reps)
            achieved = b_alg / (ms * 1e-3) / 1e9
            tr = row_traffic(name, T)
            out.append({"row": name, "what": what, "frames_per_launch": frames, "launch_ms": round(ms, 4), "frames_per_s": round(frames / (ms * 1e-3), 1),
                        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                                     "algorithmic_bytes_per_frame": round(b_alg / frames), "traffic": tr["bytes"] if tr else None,
                                     **({"traffic_source": f"profiles/{tr['source']}"} if tr else {})}})
    finally:
        cleanup()
    return out


if __name__ == "__main__":
    lib = _lib.load()
    _lib.require_device()
    name = sys.argv[1]
    loops = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 201
    rows, cleanup = build_rows(lib, T)
    launch, b_alg, frames, regex, what = rows[name]
    launch(); launch()
    ms = time_launches(lib, launch, loops)
    print(f"{name}: {ms:.4f} ms per launch of {frames} frames, {b_alg / ms / 1e6:.0f} GB/s algorithmic = {b_alg / ms / 1e6 / HBM_PEAK_GBS * 100:.1f} % of 8 TB/s ({what})")
    cleanup()
