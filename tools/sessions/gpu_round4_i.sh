#!/bin/bash
# round 4, session i: A/B of s_setprio inside fft64 (build/ab/lib_fft64prio.so) on C3 and the 64 x 64 ensemble kernel, the
# vectorised merge kernel (ensemble tests, 32 x 32 rate), strip widths once more on a second box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -2
for round in 1 2 3; do
  python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 default"
  LSPIV_LIBRARY=$R/build/ab/lib_fft64prio.so python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 fft64prio"
done
LSPIV_STRIP_W=0 python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 strip 0"
python tools/ens_launch.py 64 48 1000 5 | tail -1
LSPIV_LIBRARY=$R/build/ab/lib_fft64prio.so python tools/ens_launch.py 64 48 1000 5 | tail -1 | sed 's/^/[fft64prio] /'
python tools/ens_launch.py 32 16 1000 8 | tail -1
python tools/ens_launch.py 32 16 1000 8 | tail -1
