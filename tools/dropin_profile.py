"""Where the wall time of bench.py's drop-in and lazy legs goes: cProfile of one warm run of each (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pyorc_amd import _lib, executor, filters, frames as F, plugin  # noqa: E402
from pyorc_amd.project import Projection  # noqa: E402
from pyorc_amd.synth import particle_stack, projection_maps  # noqa: E402
from tests import lazy_doubles  # noqa: E402

lib = _lib.load(); _lib.require_device()
which = sys.argv[1] if len(sys.argv) > 1 else "dropin"
T, H, W = 201, 1080, 1920
cam = particle_stack(T, H, W, seed=5)
ws, ov = (32, 32), (16, 16)
t = np.arange(T) / 30.0
if which == "dropin":
    Ho, Wo = 810, 1440
    maps = projection_maps((H, W), (Ho, Wo), tilt=0.1, seed=1)
    norm = filters.normalize(cam, 15)
    sys.modules["xarray"] = lazy_doubles

    def run():
        video = lazy_doubles.from_frames(norm, block=20, coords={"time": t})
        ortho = lazy_doubles.frames_project(video, maps, (Ho, Wo), plugin.project_hip)
        return F.get_piv(ortho, ws[0], overlap=ov, time=t, resolution=0.01)
else:
    crop = np.ascontiguousarray(cam[:, :720, :1280])
    idx = np.roll(np.arange(720 * 1280, dtype=np.int64).reshape(720, 1280), 7, axis=1).ravel()
    lazy = bench._LazyOrthoStack(crop, idx)
    depth = None if which == "lazy" else int(which.split("=")[1])

    def run():
        return F.get_piv(lazy, ws[0], overlap=ov, time=t, resolution=0.01, prefetch=depth)

run(); run()
for _ in range(3):
    t0 = time.perf_counter(); run(); print(f"wall {time.perf_counter() - t0:.4f} s", {k: v for k, v in executor.LAST_STATS.items() if k in ("load_s", "waited_s", "upload_s", "launch_s", "chunks", "depth_per_chunk")})
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# ---- a timeline of one more warm run: who did what when (ms since the start of the run) ----
import threading
from pyorc_amd import resident, velocimetry as V, piv as PIV
log, T0 = [], [0.0]
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((label, threading.current_thread().name[-6:], (t0 - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3)); return r
    setattr(obj, name, w)
wrap(V, "load_frame_chunk", "load")
wrap(resident.ResidentStack, "stage", "stage")
wrap(resident.DeviceFrames, "from_host", " h2d")
wrap(PIV, "piv_pairs", "launch")
T0[0] = time.perf_counter(); run(); end = (time.perf_counter() - T0[0]) * 1e3
for l in sorted(log, key=lambda x: x[2]):
    print(f"{l[0]:7s} {l[1]:7s} {l[2]:7.2f} -> {l[3]:7.2f}  ({l[3] - l[2]:.2f} ms)")
print(f"end {end:.2f} ms")
