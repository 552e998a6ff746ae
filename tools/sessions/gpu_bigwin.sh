#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/bigwin
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "above_64" 2>&1 | tail -5 ) > gpurun_out/bigwin/pytest.log 2>&1
( for w in "96 48 1080 1920 200" "128 64 1080 1920 200" "128 64 2160 3840 100" "100 50 1080 1920 200" "72 36 1080 1920 200" "120 60 2160 3840 100"; do set -- $w; for nf in 0 1; do if [ $nf = 1 ]; then export LSPIV_NO_FOURSTEP=1; else unset LSPIV_NO_FOURSTEP; fi; timeout 300 python bench.py --window $1 --overlap $2 --height $3 --width $4 --pairs $5 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w nofourstep=$nf', d['value'], d['config']['mvectors_per_s'], d['config']['windows_per_pair'])"; done; done ) > gpurun_out/bigwin/bigwin.log 2>&1
tail -3 gpurun_out/bigwin/pytest.log; cat gpurun_out/bigwin/bigwin.log
