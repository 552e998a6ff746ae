#!/bin/bash
# round 4, session aa: the gather on a high-priority stream -- sharding tests, bench plumbing tests, 2-rank lines with both priorities
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4aa
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_bench.py -m gpu -q --timeout 600 2>&1 | tail -4
for pr in 1 0; do
  LSPIV_GATHER_STREAM_PRIORITY=$pr LSPIV_BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --pairs 500 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['comm']; print('priority $pr', d['value'], d['ms_per_step'], c['gather_stream_priority'], c['kernel_ms_while_gather_in_flight'], c['gather_ms_overlapped'], c['exposed_comm_ms'])"
done
LSPIV_BENCH_FORCE_COMM=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['comm']; print('rccl 1 rank', d['value'], d['ms_per_step'], c['gather_stream_priority'], c['kernel_ms_while_gather_in_flight'], c['gather_ms_overlapped'], c['exposed_comm_ms'])"
