#!/bin/bash
# round 4, session j: the fast partial-sum kernel of the ensemble rescue (power-of-two widths) and the 1x flag allowance:
# correctness (tests, strict fuzz on 24 seeds), cost
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4j
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -3
export FUZZ_MODE=ensemble FUZZ_DUMP=$R/gpurun_out/r4j/dump
for s in 103 106 107 113 $(seq 401 420); do timeout 200 python tools/fuzz_modes.py $s 80 > gpurun_out/r4j/fuzz_ens_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r4j/fuzz_ens_$s.log | cut -c1-300 | tail -3; done | sort | uniq -c
unset FUZZ_MODE FUZZ_DUMP
python tools/ens_rescue_cost.py 1000 2>&1 | tail -4
