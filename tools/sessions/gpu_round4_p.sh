#!/bin/bash
# round 4, session p: the per-timestep walking kernels without the scheduling barriers between their phases (build/ab/lib_walkrelax32.so,
# lib_walkrelax64.so) against the default, interleaved; the adopted ensemble 32 x 32 kernel (relaxed) once more
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
LSPIV_LIBRARY=$R/build/ab/lib_walkrelax32.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fft32_kernel or g3_mini or chunks_cut" --timeout 300 2>&1 | tail -1
for round in 1 2 3; do
  python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 6 --tag "c2 default"
  LSPIV_LIBRARY=$R/build/ab/lib_walkrelax32.so python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 6 --tag "c2 relax"
done
for round in 1 2; do
  python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 default"
  LSPIV_LIBRARY=$R/build/ab/lib_walkrelax64.so python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 relax"
done
python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 default"
LSPIV_LIBRARY=$R/build/ab/lib_walkrelax32.so python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 relax"
python tools/ens_launch.py 32 16 1000 8 | tail -1 | cut -c1-140
python tools/ens_launch.py 24 12 1000 8 | tail -1 | cut -c1-140
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -1
