#!/bin/bash
# round 4, session s: float32 frames, 32 x 32: HBM fetch and time over strip widths
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for sw in 24 32 40 48 64; do
  LSPIV_STRIP_W=$sw timeout 200 rocprofv3 --kernel-include-regex 'piv_fft_walk' --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$sw -o x -- python $R/tools/f32_launch.py 1000 6 > /tmp/pf_$sw.log 2>&1
  python3 - /tmp/pf_$sw "f32 strip $sw" <<'PY'
import csv, sys, glob
v=[]
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'walk_kernel' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
print(sys.argv[2], 'launches', len(v), 'fetch GB (x2)', round(2*1024*sum(v)/max(len(v),1)/1e9, 3))
PY
done
cd $R
for round in 1 2; do for sw in 0 32 48; do LSPIV_STRIP_W=$sw python tools/f32_launch.py 1000 12 | tail -1 | cut -c1-120 | sed "s/^/strip $sw /"; done; done
