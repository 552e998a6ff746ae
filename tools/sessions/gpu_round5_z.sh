#!/bin/bash
# round 5, closing session: profiles first (cool GPU), the bench line, N > 1 plumbing lines (self-launched, strong, under
# torch.distributed.run, RCCL with one rank), smoke, the GPU suite, a long strict ensemble fuzz + wide-grid parity fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5z
bash tools/gpu_round5_profiles.sh r05 2>&1 | tail -24
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2> gpurun_out/r5z/bench2.err > gpurun_out/r5z/bench_2ranks_shm.json
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --strong --strong-pairs 1000 2>> gpurun_out/r5z/bench2.err > gpurun_out/r5z/bench_2ranks_shm_strong.json
LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2>> gpurun_out/r5z/bench2.err > gpurun_out/r5z/bench_torchrun_2ranks_shm.json
LSPIV_BENCH_FORCE_COMM=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --no-extras --sustained-s 0 2>> gpurun_out/r5z/bench2.err > gpurun_out/r5z/bench_rccl_1rank.json
python - <<'PY'
import json
for f in ("bench_2ranks_shm", "bench_2ranks_shm_strong", "bench_torchrun_2ranks_shm", "bench_rccl_1rank"):
    try:
        d = json.load(open(f"gpurun_out/r5z/{f}.json")); c = d["config"]["comm"]
        print(f, d["value"], d["ms_per_step"], d["scaling"], {k: c.get(k) for k in ("transport", "mode", "pairs_total", "allgather_matches_single_launch", "kernel_ms_while_gather_in_flight", "gather_ms_overlapped", "exposed_comm_ms", "allgather_ms_alone", "kernel_ms_alone", "rccl_env")})
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 120 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -4
export FUZZ_MODE=ensemble FUZZ_DUMP=$R/gpurun_out/r5z/dump
for s in $(seq 601 616); do timeout 200 python tools/fuzz_modes.py $s 80 > gpurun_out/r5z/fuzz_ens_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r5z/fuzz_ens_$s.log | cut -c1-300 | tail -3; done | sort | uniq -c | sort -rn | head -12
grep -h "exact ties set aside" gpurun_out/r5z/fuzz_ens_*.log | grep -v " 0 exact ties" | wc -l
unset FUZZ_MODE
for s in 701 702 703 704; do FUZZ_WIDE=1 timeout 300 python tools/fuzz_parity.py $s 60 2>&1 | tail -1; done
for s in 711 712; do timeout 300 python tools/fuzz_rows.py $s 60 2>&1 | tail -1; done
