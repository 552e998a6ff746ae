#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r2f
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_filters.py -m gpu -x -q -k "above_64 or semantics or error_mapping or device_stacks" 2>&1 | tail -8 ) > gpurun_out/r2f/pytest.log 2>&1
( for w in "96 48 1080 1920 200" "128 64 1080 1920 100" "128 64 2160 3840 50" "100 50 1080 1920 100" "72 36 1080 1920 200"; do set -- $w; timeout 300 python bench.py --window $1 --overlap $2 --height $3 --width $4 --pairs $5 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['config']['mvectors_per_s'], d['config']['windows_per_pair'])"; done ) > gpurun_out/r2f/bigwin.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
tail -3 gpurun_out/r2f/pytest.log; cat gpurun_out/r2f/bigwin.log; python -c "
import json; d=json.load(open('gpurun_out/r2f/bench.json')); c=d['config']; print(d['value'], c['host_fed_pairs_per_s'], c['camera_to_velocity_pairs_per_s'])"
