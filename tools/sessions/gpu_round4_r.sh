#!/bin/bash
# round 4, session r: HBM fetch of the float32-frame C2 kernel under the strip job order (LSPIV_STRIP_W, run-time) -- time was measured
# before (no change); do the bytes move?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for sw in 0 8 16 32; do
  LSPIV_STRIP_W=$sw timeout 200 rocprofv3 --kernel-include-regex 'piv_fft_walk' --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$sw -o x -- python $R/tools/f32_launch.py 1000 6 > /tmp/pf_$sw.log 2>&1
  python3 - /tmp/pf_$sw "f32 strip $sw" <<'PY'
import csv, sys, glob
v=[]
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'walk_kernel' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print(sys.argv[2], 'launches', len(v), 'fetch GB (x2)', round(2*1024*sum(v)/max(len(v),1)/1e9, 3))
PY
  tail -1 /tmp/pf_$sw.log | cut -c1-160
done
for sw in 0 16; do
  LSPIV_STRIP_W=$sw timeout 200 rocprofv3 --kernel-include-regex 'piv_fft_walk' --pmc FETCH_SIZE --output-format csv -d /tmp/pu_$sw -o x -- python $R/tools/ab_time.py --window 32 --overlap 16 --pairs 1000 --reps 4 --warm 2 > /tmp/pu_$sw.log 2>&1
  python3 - /tmp/pu_$sw "u8 strip $sw" <<'PY'
import csv, sys, glob
v=[]
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'walk_kernel' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print(sys.argv[2], 'launches', len(v), 'fetch GB (x2)', round(2*1024*sum(v)/max(len(v),1)/1e9, 3))
PY
done
