#!/bin/bash
# round 4, session e: the 64 x 64 ensemble kernel with lane-major slots + no-return atomics (correctness, then its rate), and the
# A/B of the same scheme at 32 x 32 (three waves, no register accumulator: build/ab/lib_ens32atomic.so) against the default
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -4
FUZZ_MODE=ensemble timeout 300 python tools/fuzz_modes.py 301 80 | grep -E "FAIL|cases,"
LSPIV_LIBRARY=$R/build/ab/lib_ens32atomic.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -3
for round in 1 2 3; do
  python tools/ens_launch.py 32 16 1000 8 | tail -1
  LSPIV_LIBRARY=$R/build/ab/lib_ens32atomic.so python tools/ens_launch.py 32 16 1000 8 | tail -1 | sed 's/^/  [atomic32] /'
done
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 64 48 1000 5 | tail -1
python tools/ens_launch.py 24 12 1000 5 | tail -1
