#!/bin/bash
# instruction-fetch counters of the C2 walking kernel (is the unrolled 17 KB loop body fetch-bound?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ifetch
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 5 --cpu-pairs 0 --no-extras"
KF='--kernel-include-regex piv_fft_walk'
run() { name=$1; shift; timeout 300 rocprofv3 $KF --pmc "$@" --output-format csv -d /tmp/prof_$name -o $name -- $CMD > $OUT/$name.log 2>&1; \
        find /tmp/prof_$name -name "*counter_collection.csv" -size -8M -exec cp {} $OUT/ \; ; }
run if1 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH
run if2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READYB
run if3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
for f in $OUT/*counter_collection.csv; do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()): print(f"{c:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
done
grep -l "error code\|Error" $OUT/*.log
