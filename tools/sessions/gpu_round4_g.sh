#!/bin/bash
# round 4, session g: strip job order of the 64 x 64 walking kernels (LSPIV_STRIP_W, run-time) and the ensemble kernel after the
# epilogue reordering: correctness, rates, HBM bytes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "ensemble or fft32_kernel or config3 or chunks_cut or g3_mini" --timeout 400 2>&1 | tail -3
for round in 1 2; do
  for sw in 0 32 16 64; do LSPIV_STRIP_W=$sw python tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 4 --tag "c3 strip $sw"; done
done
for sw in 0 32; do LSPIV_STRIP_W=$sw python tools/ens_launch.py 64 48 1000 5 | tail -1 | sed "s/^/strip $sw /"; done
LSPIV_STRIP_W=16 python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --tag "c2 strip 16"
python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --tag "c2 strip 0"
LSPIV_STRIP_W=16 python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 strip 16"
LSPIV_STRIP_W=32 python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 strip 32"
python tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 5 --dtype f32 --tag "c2 f32 strip 0"
# HBM bytes of the 64 x 64 walking kernel with and without strips
cd /tmp && export TMPDIR=/tmp
for sw in 0 32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    LSPIV_STRIP_W=$sw timeout 200 rocprofv3 --kernel-include-regex 'piv_fft_walk' --pmc $c --output-format csv -d /tmp/p_$sw_$c -o x -- python $R/tools/ab_time.py --window 64 --overlap 48 --pairs 1000 --reps 3 --warm 2 > /dev/null 2>&1
    python3 - /tmp/p_$sw_$c "strip $sw $c" <<'PY'
import csv, sys, glob
v=[]
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'walk_kernel' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print(sys.argv[2], 'launches', len(v), 'mean KiB', sum(v)/max(len(v),1))
PY
    rm -rf /tmp/p_$sw_$c
  done
done
