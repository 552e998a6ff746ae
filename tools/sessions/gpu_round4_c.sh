#!/bin/bash
# round 4, session c: first run of the ensemble rescue + ADVICE fixes: the new tests, the strict ensemble fuzz on the seeds that
# failed in session b and on fresh ones, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble_rescue or many_way_tie" --timeout 300 2>&1 | tail -30 > gpurun_out/r4c/new_tests.log
tail -30 gpurun_out/r4c/new_tests.log
export FUZZ_MODE=ensemble FUZZ_DUMP=$R/gpurun_out/r4c/dump
for s in 103 106 107 113 $(seq 201 212); do timeout 200 python tools/fuzz_modes.py $s 80 > gpurun_out/r4c/fuzz_ens_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r4c/fuzz_ens_$s.log | cut -c1-300 | tail -4; done
unset FUZZ_MODE FUZZ_DUMP
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -15 > gpurun_out/r4c/suite.log
tail -15 gpurun_out/r4c/suite.log
