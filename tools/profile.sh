#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun).  Trace and counter passes
# are separate runs (the pool refuses --pmc combined with trace domains other than kernel-trace/stats).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-extras --sustained-s 0 ${BENCH_ARGS}"
[ -n "$PROFILE_CMD" ] && CMD="$PROFILE_CMD"   # another workload through the same passes (tools/f32_launch.py)
KF=${PROFILE_KF:-'--kernel-include-regex piv_'}   # PROFILE_KF: the counter passes of another kernel family (tools/rows_launch.py)
run() { name=$1; shift; timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o $name -- $CMD > $OUT/$name.log 2>&1; \
        find /tmp/prof_$name -name "*.csv" -size -8M -exec cp {} $OUT/ \; ; }
# the trace pass runs 100 timed steps so that the cold first launches do not weigh on the kernel's mean duration
CMD_SAVE=$CMD; [ -z "$PROFILE_CMD" ] && CMD="python $R/bench.py --steps 100 --warmup 3 --cpu-pairs 0 --no-extras --sustained-s 0 ${BENCH_ARGS}"
run trace --kernel-trace --stats
CMD=$CMD_SAVE
run pmc_fetch $KF --pmc FETCH_SIZE
run pmc_write $KF --pmc WRITE_SIZE
run pmc_sq1 $KF --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run pmc_sq2 $KF --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run pmc_grbm $KF --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_tcc $KF --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
ls -la $OUT
head -5 $OUT/trace_kernel_stats.csv
for f in $OUT/pmc_*counter_collection.csv; do echo == $f; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
done
tail -2 $OUT/trace.log
