#!/bin/bash
# Round 6 closing: bench.py as the driver launches it for N = 2 (torch.distributed.run), the two ranks sharing this box's one GPU
cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/bench2; mkdir -p $OUT
LSPIV_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_torchrun2.json 2> $OUT/bench_torchrun2.err
echo "rc $?"; tail -3 $OUT/bench_torchrun2.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench2/bench_torchrun2.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, json.dumps(d["config"].get("comm"))[:700])
PY
