#!/bin/bash
# round 4, session w: the new full-size tests (ensemble on full-width grids; the device chain beyond 2^31 elements)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -k "full_width or beyond" --durations=5 2>&1 | tail -40
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4w_bench.err > gpurun_out/r4w_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4w_bench.json"))
print(d["value"], d["ms_per_step"]); print(d["config"].get("camera_to_velocity_pairs_per_s")); print(d["config"].get("host_fed_pairs_per_s"))
PY
