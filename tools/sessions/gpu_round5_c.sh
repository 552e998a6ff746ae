#!/bin/bash
# round 5, session c: the GPU tests session b did not reach (it stopped at the first failure: get_piv turned a lazy stack into one array),
# then the bench line with its new legs (kernel timed inside the launch, sustained loop, configs[2] / [3] parity, lazy host chunks)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5c
LSPIV_TIE_LOG=$R/gpurun_out/r5c/ties.log timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_shard.py tests/test_gpu_strip_order.py tests/test_masks.py tests/test_project.py tests/test_filters.py tests/test_c_example.py -m gpu -q --timeout 600 2>&1 | tail -15
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5c/bench.err > gpurun_out/r5c/bench.json; echo "bench rc $?"; tail -3 gpurun_out/r5c/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c/bench.json"))
print(d["value"], d["ms_per_step"], d.get("sustained_pairs_per_s"), d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["roofline"].get("kernel_ms_same_launch_rescue_off"))
print(d["config"].get("sustained"))
for o in d["config"].get("other_configs", []):
    print(o["workload"][:60], o["pairs_per_s"], o.get("kernel_ms"), o.get("parity_vs_oracle"), o["roofline"].get("traffic"))
print(d["config"].get("lazy_host_chunks"))
print(d["config"].get("host_fed_pairs_per_s"))
print({k: v for k, v in d["cpu_baseline"].items() if k.startswith("parity") and k != "parity_float64_ties"}, d["cpu_baseline"]["value"], d["cpu_baseline"].get("ensemble_parity"))
PY
