#!/bin/bash
# round 3: kernel-trace of the PIV kernel + the two rescue kernels (C2 and C3)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "32 16 10" "64 48 4"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t$1 -- python $R/tools/ab_time.py --window $1 --overlap $2 --reps $3 > $OUT/t$1.log 2>&1
  find /tmp/prof_$1 -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
  tail -1 $OUT/t$1.log
  python3 - $OUT/t$1_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:80], r["Calls"], "avg_us", float(r["AverageNs"])/1e3, "min", float(r["MinNs"])/1e3, "max", float(r["MaxNs"])/1e3)
PY
done
