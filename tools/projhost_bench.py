import sys, time, numpy as np
sys.path.insert(0, '.')
from pyorc_amd.project import Projection
from pyorc_amd.synth import particle_stack, projection_maps
src, dst = (1080, 1920), (810, 1440)
p = Projection(src, dst, *projection_maps(src, dst, tilt=0.1, seed=1))
cam = particle_stack(20, *src, seed=1)
p.project_frames(cam); p.project_frames(cam)
t0 = time.perf_counter()
for _ in range(10): out = p.project_frames(cam)
dt = (time.perf_counter() - t0) / 10
print(f"host project_frames 20 frames: {dt*1e3:.2f} ms = {20/dt:.0f} frames/s ({(cam.nbytes + out.nbytes)/dt/1e9:.1f} GB/s over PCIe both ways)")
