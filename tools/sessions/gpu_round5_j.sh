#!/bin/bash
# round 5, session j: HBM fetch of the C2 / C3 / ensemble kernels against the strip width of the job order, under the new defaults
# (XCD partition by windows, 125-pair anchors)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
prof() {  # label, command...
  label=$1; shift
  rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-include-regex piv_ --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- "$@" > /tmp/pf.log 2>&1
  python3 - "$label" <<'PY'
import csv, glob, sys
v = []
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "walk" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": v.append(float(r["Counter_Value"]))
print(sys.argv[1], "fetch GB per launch", round(2 * 1024 * sum(v) / max(len(v), 1) / 1e9, 3))
PY
}
for sw in 0 8 16 24 32 48; do
  export LSPIV_STRIP_W=$sw
  LSPIV_RESCUE=0 prof "c2 strip $sw" python $R/tools/ab_time.py --window 32 --overlap 16 --reps 3 --warm 2
  LSPIV_RESCUE=0 python $R/tools/ab_time.py --window 32 --overlap 16 --tag "c2 strip $sw" | tail -1
done
for sw in 0 8 16 32 64; do
  export LSPIV_STRIP_W=$sw
  LSPIV_RESCUE=0 prof "c3 strip $sw" python $R/tools/ab_time.py --window 64 --overlap 48 --reps 3 --warm 2
  LSPIV_RESCUE=0 python $R/tools/ab_time.py --window 64 --overlap 48 --tag "c3 strip $sw" | tail -1
done
