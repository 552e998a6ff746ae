#!/bin/bash
# Round 6: what the rescue pass costs on the recipe's own frames (edge-detected, clipped, projected float32): kernel durations and the
# gaps between the kernels of back-to-back launches, from the kernel trace's timestamps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for resc in 1 0; do
LSPIV_RESCUE=$resc PROBE_ONE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_probe$resc -o probe -- python $R/tools/recipe_piv_probe.py ${PAIRS:-200} > /dev/null 2>&1
f=$(find /tmp/prof_probe$resc -name "*kernel_trace.csv" | head -1); echo "LSPIV_RESCUE=$resc"; python3 - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "piv_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-30:]
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].split("::")[-1][:28]
    print(f"{name:30s} dur {(e-s)/1e3:8.1f} us   gap before {((s-prev)/1e3 if prev else 0):7.1f} us")
    prev=e
PY
done
