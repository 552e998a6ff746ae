import os, sys
sys.path.insert(0, '/root/repo')
os.environ["LSPIV_PROJECT_DEBUG"] = "1"
from pyorc_amd.synth import projection_maps
from pyorc_amd.project import Projection
for src, dst, tilt, seed in [((540, 960), (200, 360), 0.3, 3), ((540, 960), (200, 360), 0.3, 5), ((270, 480), (200, 360), 0.9, 3), ((270, 480), (200, 360), 0.9, 5), ((405, 720), (200, 360), 0.3, 5), ((272, 488), (120, 520), 0.2, 5), ((270, 480), (200, 358), 0.35, 5)]:
    print(src, dst, tilt, seed, flush=True)
    p = Projection(src, dst, *projection_maps(src, dst, tilt=tilt, seed=seed)); p.close()
