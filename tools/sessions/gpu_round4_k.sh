#!/bin/bash
# round 4, session k: 64 x 64 ensemble kernel with SGPR-base slot addressing (written-out stores / LDS-DMA): correctness, rate, bytes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -2
FUZZ_MODE=ensemble timeout 300 python tools/fuzz_modes.py 301 80 | grep -E "FAIL|cases,"
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 32 16 1000 8 | tail -1
python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3"
python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --tag "c2"
bash tools/profile_ens.sh r04k 64 48 2>&1 | grep -E "walk_ensemble|merge" | cut -c1-200
