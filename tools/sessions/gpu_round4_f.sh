#!/bin/bash
# round 4, session f: 64 x 64 ensemble kernel, partial sum fetched into the idle transpose tile (global_load_lds) + coalesced stores
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -3
FUZZ_MODE=ensemble timeout 300 python tools/fuzz_modes.py 301 80 | grep -E "FAIL|cases,"
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 32 16 1000 8 | tail -1
python tools/ens_launch.py 48 24 1000 5 | tail -1
