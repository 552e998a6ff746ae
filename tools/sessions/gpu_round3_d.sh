#!/bin/bash
# round 3: GPU suite + the cost of the RCCL gather with one rank
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3d; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -5
for round in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'])"
  LSPIV_BENCH_FORCE_COMM=1 timeout 200 python bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-extras 2>$OUT/comm.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rccl1', d['value'], d['ms_per_step'], d['config'].get('comm'))"
done
cd /tmp && export TMPDIR=/tmp
LSPIV_BENCH_FORCE_COMM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_comm -o comm -- python $R/bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-extras > $OUT/trace.log 2>&1
find /tmp/prof_comm -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
find /tmp/prof_comm -name "*kernel_trace.csv" -exec cp {} $OUT/ \;
python3 - $OUT/comm_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 1e5 and "synth" not in r["Name"]:
        print(r["Name"][:90], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3, 1), "total_ms", round(float(r["TotalDurationNs"])/1e6, 2))
PY
python3 - $OUT/comm_kernel_trace.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "synth" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-24:]:
    print(r["Kernel_Name"][:60], "stream", r.get("Stream_Id", r.get("Queue_Id")), "start_us", (int(r["Start_Timestamp"]) - t0) / 1e3, "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
