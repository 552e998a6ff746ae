"""Throughput of project_cv (cv2.undistort + cv2.warpPerspective restated as two fixed-point bilinear remaps) on an
HBM-resident 1080p stack: frames/s and algorithmic GB/s (source frame read once, destination written once, per remap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd import DeviceFrames, _lib
from pyorc_amd.project import ProjectionCV
lib = _lib.load(); _lib.require_device()
T, H, W = 401, 1080, 1920
K = np.array([[1500.0, 0, 960], [0, 1500.0, 540], [0, 0, 1]])
dist = [-0.12, 0.03, 0.0005, -0.0003, 0.0]
M = np.array([[0.95, 0.04, 20.0], [0.01, 0.9, 30.0], [1e-5, 2e-5, 1.0]])
for dtype in (np.uint8, np.float32):
    d = DeviceFrames.empty((T, H, W), np.uint8)
    _lib.check(lib.lspiv_synth_particles_dev(d.c_ptr, T, H, W, 3, 0.02))
    if dtype == np.float32:
        from pyorc_amd import filters
        d = filters.smooth(d, 1)
    for name, k, dc in (("undistort + warp", K, dist), ("warp only", None, None)):
        p = ProjectionCV((H, W), (H, W), k, dc, M)
        p.project_frames(d); _lib.check(lib.lspiv_synchronize())
        t0 = time.perf_counter()
        for _ in range(3): o = p.project_frames(d)
        _lib.check(lib.lspiv_synchronize()); t = (time.perf_counter() - t0) / 3
        b = np.dtype(dtype).itemsize * H * W * (4 if k is not None else 2)
        print(f"project_cv {np.dtype(dtype).name} {name}: {T/t:.0f} frames/s, {T*b/t/1e12:.2f} TB/s algorithmic ({b/1e6:.1f} MB/frame)", flush=True)
        p.close()
