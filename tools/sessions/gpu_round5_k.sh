#!/bin/bash
# round 5, session k: anchor length under the XCD partition by windows -- time and HBM fetch of C2 / C3 / ensemble 64 / ensemble 32 (1000 pairs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
fetch() {  # label, command...
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
for w in 25 51 75 101 125 175; do
  export LSPIV_WALK=$w
  LSPIV_RESCUE=0 python $R/tools/ab_time.py --window 32 --overlap 16 --tag "c2 anchor $w" | tail -1
  LSPIV_RESCUE=0 fetch "c2 anchor $w" python $R/tools/ab_time.py --window 32 --overlap 16 --reps 3 --warm 2
  LSPIV_RESCUE=0 python $R/tools/ab_time.py --window 64 --overlap 48 --tag "c3 anchor $w" | tail -1
  LSPIV_RESCUE=0 fetch "c3 anchor $w" python $R/tools/ab_time.py --window 64 --overlap 48 --reps 3 --warm 2
  python $R/tools/ens_launch.py 64 48 1000 4 | cut -c88-140 | sed "s/^/ens64 anchor $w: /"
  python $R/tools/ens_launch.py 32 16 1000 6 | cut -c88-140 | sed "s/^/ens32 anchor $w: /"
done
