#!/bin/bash
# round 4, session b: long strict caller fuzz, ensemble mode only (gate: EVERY window within 1e-4), failing cases dumped
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4b
export FUZZ_MODE=ensemble FUZZ_DUMP=$R/gpurun_out/r4b/dump
for s in $(seq 101 116); do timeout 200 python tools/fuzz_modes.py $s 80 > gpurun_out/r4b/fuzz_ens_$s.log 2>&1; grep -E "FAIL|cases," gpurun_out/r4b/fuzz_ens_$s.log | cut -c1-400 | tail -6; done
du -sh gpurun_out/r4b/dump 2>/dev/null
