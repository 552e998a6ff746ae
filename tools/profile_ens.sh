#!/bin/bash
# rocprofv3 passes of the ensemble kernels (tools/ens_launch.py): trace, HBM bytes, L2, wave statistics -> gpurun_out/prof_<tag>_ens<N>
# usage: profile_ens.sh <tag> [window overlap]...   (default: 32 16 and 64 48)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}; shift
[ $# -eq 0 ] && set -- 32 16 64 48
while [ $# -ge 2 ]; do
  WS=$1; OV=$2; shift 2
  OUT=$R/gpurun_out/prof_${TAG}_ens$WS; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
    CMD="python $R/tools/ens_launch.py $WS $OV 1000 6"
    KF='--kernel-include-regex piv_|ensemble'
    run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o $name -- $CMD > $OUT/$name.log 2>&1; find /tmp/prof_$name -name "*.csv" -size -8M -exec cp {} $OUT/ \; ; rm -rf /tmp/prof_$name; }
    run trace --kernel-trace --stats
    run pmc_fetch $KF --pmc FETCH_SIZE
    run pmc_write $KF --pmc WRITE_SIZE
    run pmc_tcc $KF --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    run pmc_sq1 $KF --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    run pmc_grbm $KF --pmc GRBM_GUI_ACTIVE GRBM_COUNT
  )
  grep -E "piv_|ensemble_m" $OUT/trace_kernel_stats.csv | cut -c1-220; tail -1 $OUT/trace.log
  for f in $OUT/pmc_*counter_collection.csv; do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "walk_ensemble" in k: print(f"{k:50s} {c:22s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
  done
done
