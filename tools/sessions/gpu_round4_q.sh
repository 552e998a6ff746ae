#!/bin/bash
# round 4, session q: the relaxed 32 x 32 ensemble kernel per variant (uint8 / float32, with and without a signal threshold)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -1
FUZZ_MODE=ensemble timeout 300 python tools/fuzz_modes.py 601 80 | grep -E "FAIL|cases,"
for v in "u8 -1" "u8 0.05" "f32 -1" "f32 0.05"; do python tools/ens_launch.py 32 16 1000 8 $v | tail -1 | cut -c1-170; done
python tools/ens_launch.py 64 48 1000 5 | tail -1 | cut -c1-170
