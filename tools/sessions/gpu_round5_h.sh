#!/bin/bash
# round 5, session h: XCD partition by windows + 125-pair anchors on large grids as the defaults: the whole GPU suite, fuzz, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5h
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8
for s in 901 902; do FUZZ_WIDE=1 timeout 300 python tools/fuzz_parity.py $s 60 2>&1 | tail -1; done
for s in 911 912; do timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1; done
for s in 921 922; do FUZZ_MODE=ensemble timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5h/bench.err > gpurun_out/r5h/bench.json; echo "bench rc $?"; tail -2 gpurun_out/r5h/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5h/bench.json"))
print(d["value"], d["ms_per_step"], d.get("sustained_pairs_per_s"), d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"])
for o in d["config"].get("other_configs", []):
    print(o["workload"][:60], o["pairs_per_s"], o.get("kernel_ms"), o.get("parity_vs_oracle", {}).get("max_rel_err_vs_oracle"), o.get("finish_ms"))
c = d["cpu_baseline"]; print({k: v for k, v in c.items() if k.startswith("parity") and not isinstance(v, dict)}, c.get("ensemble_parity"))
PY
