#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2e
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2e/pytest.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
( timeout 300 python tools/ensemble_bench.py 501 ) > gpurun_out/r2e/ensemble.log 2>&1
( timeout 300 python tools/project_cv_bench.py ) > gpurun_out/r2e/project_cv.log 2>&1
( for w in "96 48 1080 1920 200" "128 64 1080 1920 100" "128 64 2160 3840 50" "100 50 1080 1920 100"; do set -- $w; timeout 300 python bench.py --window $1 --overlap $2 --height $3 --width $4 --pairs $5 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['config']['mvectors_per_s'], d['config']['windows_per_pair'])"; done ) > gpurun_out/r2e/bigwin.log 2>&1
tail -3 gpurun_out/r2e/pytest.log; cat gpurun_out/r2e/ensemble.log gpurun_out/r2e/project_cv.log gpurun_out/r2e/bigwin.log; python -c "
import json; d=json.load(open('gpurun_out/r2e/bench.json')); c=d['config']; print(d['value'], c['host_fed_pairs_per_s'], c['camera_to_velocity_pairs_per_s'], [o['pairs_per_s'] for o in c['other_configs']])"
