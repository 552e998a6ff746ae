#!/bin/bash
# round 3: the uint8-staying projection (nearest-neighbour-only plans): tests, then project + get_piv on HBM-resident stacks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r3l
timeout 600 python -m pytest tests/test_project.py tests/test_filters.py tests/test_host.py -m gpu -q --timeout 300 2>&1 | tail -5
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r3l/chain.json
import json, sys
sys.path.insert(0, '.')
import bench
from pyorc_amd.synth import particle_stack
cam = particle_stack(201, 1080, 1920, seed=3, density=0.02)
print(json.dumps(bench.camera_to_velocity_rates(cam, (32, 32), (16, 16))))
PY
timeout 200 python tools/project_bench.py 2>&1 | tail -5
