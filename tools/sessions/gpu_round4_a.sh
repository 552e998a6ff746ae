#!/bin/bash
# round 4, session a (diagnostic, on the round-3 kernels): which ensemble / > 64 px cases miss the 1e-4 gate and why
# (tools/ens_diag.py), the strict caller fuzz, and a first counter set of the ensemble kernels (trace + HBM bytes + L2)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4a
timeout 400 python tools/ens_diag.py 1e-4 > gpurun_out/r4a/ens_diag.log 2>&1
tail -60 gpurun_out/r4a/ens_diag.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "windows_above or get_ffpiv_ensemble_vs_oracle or ensemble_other_window_size" --timeout 200 2>&1 | tail -40 > gpurun_out/r4a/tests_tol.log
tail -25 gpurun_out/r4a/tests_tol.log
for s in 21 22; do timeout 300 python tools/fuzz_modes.py $s 60 > gpurun_out/r4a/fuzz_modes_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r4a/fuzz_modes_$s.log | tail -15; done
# ensemble kernels: trace + traffic
export PROFILE_PASSES_SHORT=1
for cfg in "32 16" "64 48"; do set -- $cfg
  OUT=$R/gpurun_out/prof_r04a_ens$1; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
    CMD="python $R/tools/ens_launch.py $1 $2 1000 6"
    KF='--kernel-include-regex piv_|ensemble'
    run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o $name -- $CMD > $OUT/$name.log 2>&1; find /tmp/prof_$name -name "*.csv" -size -8M -exec cp {} $OUT/ \; ; rm -rf /tmp/prof_$name; }
    run trace --kernel-trace --stats
    run pmc_fetch $KF --pmc FETCH_SIZE
    run pmc_write $KF --pmc WRITE_SIZE
    run pmc_tcc $KF --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    run pmc_sq1 $KF --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  )
  head -6 $OUT/trace_kernel_stats.csv; tail -1 $OUT/trace.log
  for f in $OUT/pmc_*counter_collection.csv; do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:50s} {c:22s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
  done
done
