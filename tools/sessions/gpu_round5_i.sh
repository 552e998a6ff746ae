#!/bin/bash
# round 5, session i: where does the extra HBM fetch of the new defaults come from -- the XCD partition by windows or the 125-pair anchors?
# FETCH_SIZE of the C2 / C3 kernels under the four combinations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "32 16" "64 48"; do
  set -- $cfg
  for o in 0 1; do
    for w in 25 125; do
      rm -rf /tmp/pf; LSPIV_XCD_ORDER=$o LSPIV_WALK=$w LSPIV_RESCUE=0 timeout 300 rocprofv3 --kernel-include-regex piv_ --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/tools/ab_time.py --window $1 --overlap $2 --reps 3 --warm 2 > /tmp/pf.log 2>&1
      python3 - "$1 order $o anchor $w" <<'PY'
import csv, glob, sys
v = []
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": v.append(float(r["Counter_Value"]))
print(sys.argv[1], "launches", len(v), "fetch GB per launch", round(2 * 1024 * sum(v) / max(len(v), 1) / 1e9, 3))
PY
    done
  done
done
