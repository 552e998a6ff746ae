# rocprofv3 WRITE_SIZE / FETCH_SIZE of tools/ubench/l2_rmw for: mode (0 RMW alone, 1 + plain stream, 2 + non-temporal stream) x layout
# (0 contiguous hot bytes, 1 the hot 8 KB of every 16 KB) x period (stream intensity)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/l2_rmw; rm -f $R/gpurun_out/l2_rmw/*.log
for cfg in ${RMW_CFGS:-"0 0 1 0" "0 1 1 0" "1 0 1 0" "1 1 1 0" "1 0 8 0" "1 1 8 0" "1 0 32 0" "1 1 32 0" "2 0 1 0" "2 1 8 0" "0 0 1 4" "1 0 32 4"}; do
  set -- $cfg
  $R/tools/ubench/l2_rmw $1 500 $2 $3 $4 >> $R/gpurun_out/l2_rmw/run.log
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/rmw_$1$2$3$4_$c -o p -- $R/tools/ubench/l2_rmw $1 500 $2 $3 $4 > /dev/null 2>&1
    f=$(find /tmp/rmw_$1$2$3$4_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$cfg" >> $R/gpurun_out/l2_rmw/counters.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "rmw" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(f"mode/layout/period/naps {sys.argv[2]}: {k} {v * 1024 / 1e9:.3f} GB")
PY
  done
done
cat $R/gpurun_out/l2_rmw/run.log $R/gpurun_out/l2_rmw/counters.log
