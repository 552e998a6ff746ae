#!/bin/bash
# round 4, session ae: clock and socket power while each of the four walking kernels runs for ~8 s (the ensemble kernels take as many
# CYCLES as their per-timestep twins or fewer -- profiles/r04_*: GRBM_GUI_ACTIVE -- but more time: what clock do they get?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4ae
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'sclk\|mclk\|Package Power' | tr -s ' \t' ' ' | tr '\n' '|')"; sleep 0.5; done ) >> gpurun_out/r4ae/smi.log 2>&1 &   # append mode: the markers below go to the same file
SMI=$!
run() { echo "$(date +%s.%N) start $1" >> gpurun_out/r4ae/smi.log; shift; "$@" | tail -1 | cut -c1-150; echo "$(date +%s.%N) stop" >> gpurun_out/r4ae/smi.log; }
run c2    env LSPIV_RESCUE=0 python tools/ab_time.py --window 32 --overlap 16 --reps 1300 --tag c2
run ens32 python tools/ens_launch.py 32 16 1000 1300
run c3    env LSPIV_RESCUE=0 python tools/ab_time.py --window 64 --overlap 48 --reps 280 --tag c3
run ens64 python tools/ens_launch.py 64 48 1000 250
kill $SMI
python3 - <<'PY'
import re, statistics
cur, data = None, {}
for line in open("gpurun_out/r4ae/smi.log"):
    m = re.match(r"[\d.]+ start (\w+)", line)
    if m: cur = m.group(1); data[cur] = []; continue
    if " stop" in line: cur = None; continue
    if cur:
        s = re.search(r"sclk[^(]*\((\d+)Mhz\)", line); p = re.search(r"Power \(W\): ([\d.]+)", line)
        if s and p: data[cur].append((int(s.group(1)), float(p.group(1))))
for k, v in data.items():
    v = v[len(v) // 2:]          # the second half of the run: past the load phase (stack synthesis, warm-up)
    if v: print(k, "samples", len(v), "sclk median", statistics.median(x[0] for x in v), "MHz  power median", statistics.median(x[1] for x in v), "W")
PY
