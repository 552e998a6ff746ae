"""How long lspiv_projection_create takes at 1080p -> 810 x 1440 (three plans since round 6: windows, mixed + tiles, float32 tiles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyorc_amd.synth import projection_maps
from pyorc_amd.project import Projection
maps = projection_maps((1080, 1920), (810, 1440), tilt=0.1, seed=1)
for env in ({}, {"LSPIV_PROJECT_NO_TILE": "1"}, {"LSPIV_PROJECT_NO_TILE": "1", "LSPIV_PROJECT_NO_MIX": "1"}):
    for k in ("LSPIV_PROJECT_NO_TILE", "LSPIV_PROJECT_NO_MIX"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(2):
        t0 = time.perf_counter(); p = Projection((1080, 1920), (810, 1440), *maps); dt = time.perf_counter() - t0; p.close()
    print(env or "default", f"{dt * 1e3:.0f} ms")
