#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/bigwin
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "above_64" 2>&1 | tail -5 ) > gpurun_out/bigwin/pytest.log 2>&1
( for w in "96 48 1080 1920 200" "128 64 1080 1920 100" "128 64 2160 3840 50" "100 50 1080 1920 100" "72 36 1080 1920 200"; do set -- $w; timeout 300 python bench.py --window $1 --overlap $2 --height $3 --width $4 --pairs $5 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['config']['mvectors_per_s'], d['config']['windows_per_pair'])"; done ) > gpurun_out/bigwin/bigwin.log 2>&1
tail -3 gpurun_out/bigwin/pytest.log; cat gpurun_out/bigwin/bigwin.log
