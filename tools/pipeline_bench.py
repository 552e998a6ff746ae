"""End-to-end: raw uint8 camera frames in host memory -> velocities in host memory, every stage on the GPU
(pipeline.CameraToVelocity).  PCIe-inclusive; compares with the host-fed PIV-only rate and the CPU oracle chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyorc_amd.pipeline import CameraToVelocity
from pyorc_amd.synth import projection_maps

src, dst, T = (1080, 1920), (1080, 1920), int(sys.argv[1]) if len(sys.argv) > 1 else 201
maps = projection_maps(src, dst, tilt=0.3, seed=1)
rng = np.random.default_rng(0)
base = (rng.random(src) ** 6 * 255).astype(np.uint8)
cam = np.stack([np.roll(base, (2 * t, -5 * t), (0, 1)) for t in range(T)])
for samples, packed, edge, ws in ((None, False, None, 32), (15, False, None, 32), (15, True, None, 32), (15, True, (1, 2), 32), (15, False, None, 64),
                                  (15, False, (1, 2), 64)):
    with CameraToVelocity(src, dst, *maps, window_size=(ws, ws), overlap=(ws // 2 if ws == 32 else 48,) * 2, normalize_samples=samples,
                          edge_detect=edge, minmax=(-5, 5) if edge else None) as chain:
        chain.run(cam[:31])
        rate = {}
        for streamed in (False, True):      # one piece | upload in time chunks under the kernels of the previous chunk
            chain.run(cam, packed=packed, streamed=streamed)
            best = min((lambda t0: (chain.run(cam, packed=packed, streamed=streamed), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
            rate[streamed] = best
    print(f"window {ws} normalize={samples} edge_detect={edge} packed={packed}: {T-1} pairs in {rate[False]*1e3:.1f} ms one piece -> {(T-1)/rate[False]:.0f} pairs/s "
          f"host-to-host ({cam.nbytes/rate[False]/1e9:.1f} GB/s of camera frames); streamed {rate[True]*1e3:.1f} ms -> {(T-1)/rate[True]:.0f} pairs/s", flush=True)
