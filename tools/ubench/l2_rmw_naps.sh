cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
run() { # mode layout period naps
  $R/tools/ubench/l2_rmw $1 500 $2 $3 $4 | cut -c1-110
  timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/n_$1$2$3$4 -o p -- $R/tools/ubench/l2_rmw $1 500 $2 $3 $4 > /dev/null 2>&1
  f=$(find /tmp/n_$1$2$3$4 -name "*counter_collection.csv" | head -1)
  python3 -c "
import csv,sys
print('   WRITE_SIZE GB', sum(float(r['Counter_Value']) for r in csv.DictReader(open('$f')) if 'rmw' in r['Kernel_Name'])*1024/1e9)"
}
run 0 0 1 0; run 5 0 1 0; run 5 0 1 4; run 5 1 1 0
