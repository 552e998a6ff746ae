#!/bin/bash
# round 5, session b: the tree with the half-LDS 64 x 64 ensemble kernel, the chunk executor, per-device locks and the ADVICE r04 fixes:
# numerical difference of the ensemble kernels against the round-4 build, the whole GPU suite (tie shares logged), the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5b
for thr in -1 0.2; do
  ENS_DUMP=/tmp/tree_$thr.npz python tools/ens_hash.py 64 48 120 $thr | tail -1 | sed 's/^/tree /'
  ENS_DUMP=/tmp/r04_$thr.npz LSPIV_LIBRARY=$R/build/ab/lib_r04.so python tools/ens_hash.py 64 48 120 $thr | tail -1 | sed 's/^/r04  /'
  python tools/ens_diff.py /tmp/r04_$thr.npz /tmp/tree_$thr.npz
done
rm -f gpurun_out/r5b/ties.log
LSPIV_TIE_LOG=$R/gpurun_out/r5b/ties.log timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -15
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5b/bench.err > gpurun_out/r5b/bench.json; echo "bench rc $?"; tail -3 gpurun_out/r5b/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench.json"))
print(d["value"], d["ms_per_step"], d.get("sustained_pairs_per_s"), d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["roofline"].get("kernel_ms_same_launch_rescue_off"))
print(d["config"].get("sustained"))
for o in d["config"].get("other_configs", []):
    print(o["workload"][:60], o["pairs_per_s"], o.get("kernel_ms"), o.get("parity_vs_oracle"), o["roofline"].get("traffic"))
print(d["config"].get("lazy_host_chunks"))
print(d["config"].get("host_fed_pairs_per_s"))
print({k: v for k, v in d["cpu_baseline"].items() if k.startswith("parity") and k != "parity_float64_ties"}, d["cpu_baseline"]["value"], d["cpu_baseline"].get("ensemble_parity"))
PY
