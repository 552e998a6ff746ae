#!/bin/bash
# round 4, closing session: profiles first (cool GPU), the bench line, N > 1 plumbing lines (self-launched, strong, under
# torch.distributed.run), then the GPU suite and a long strict ensemble fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4l
bash tools/gpu_round4_profiles.sh r04 2>&1 | tail -22
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2> gpurun_out/r4l/bench2.err > gpurun_out/r4l/bench_2ranks_shm.json
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --strong --strong-pairs 1000 2>> gpurun_out/r4l/bench2.err > gpurun_out/r4l/bench_2ranks_shm_strong.json
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2>> gpurun_out/r4l/bench2.err > gpurun_out/r4l/bench_torchrun_2ranks_shm.json
LSPIV_BENCH_FORCE_COMM=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --no-extras 2>> gpurun_out/r4l/bench2.err > gpurun_out/r4l/bench_rccl_1rank.json
python - <<'PY'
import json
for f in ("bench_2ranks_shm", "bench_2ranks_shm_strong", "bench_torchrun_2ranks_shm", "bench_rccl_1rank"):
    try:
        d = json.load(open(f"gpurun_out/r4l/{f}.json")); c = d["config"]["comm"]
        print(f, d["value"], d["ms_per_step"], d["scaling"], {k: c.get(k) for k in ("transport", "mode", "pairs_total", "allgather_matches_single_launch", "kernel_ms_while_gather_in_flight", "gather_ms_overlapped", "exposed_comm_ms", "allgather_ms_alone", "kernel_ms_alone", "rccl_env")})
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 120 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -4
export FUZZ_MODE=ensemble FUZZ_DUMP=$R/gpurun_out/r4l/dump
for s in $(seq 501 532); do timeout 200 python tools/fuzz_modes.py $s 80 > gpurun_out/r4l/fuzz_ens_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r4l/fuzz_ens_$s.log | cut -c1-300 | tail -3; done | sort | uniq -c | sort -rn | head -12
grep -h "exact ties set aside" gpurun_out/r4l/fuzz_ens_*.log | grep -v " 0 exact ties" | wc -l
