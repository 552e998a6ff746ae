#!/bin/bash
# round 5, session n: long fuzz on the final sources (kernel hash eca1f282f075500b): kernels (wide grids and small frames), callers per time
# step and in ensemble mode (every window), the rows around the path
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5n
export FUZZ_DUMP=$R/gpurun_out/r5n/dump
for s in $(seq 1001 1008); do FUZZ_WIDE=1 timeout 300 python tools/fuzz_parity.py $s 60 2>&1 | tail -1; done | sort | uniq -c
for s in $(seq 1011 1030); do timeout 300 python tools/fuzz_parity.py $s 100 2>&1 | tail -1 | cut -c1-40; done | sort | uniq -c
for s in $(seq 1031 1046); do timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1 | cut -c1-40; done | sort | uniq -c
for s in $(seq 1051 1082); do FUZZ_MODE=ensemble timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1 | cut -c1-40; done | sort | uniq -c
for s in $(seq 1091 1098); do FUZZ_MODE=ensemble FUZZ_WIDE=1 timeout 300 python tools/fuzz_modes.py $s 40 2>&1 | grep -E "FAIL|cases," | tail -1 | cut -c1-40; done | sort | uniq -c
for s in $(seq 1101 1108); do timeout 300 python tools/fuzz_rows.py $s 60 2>&1 | tail -1 | cut -c1-40; done | sort | uniq -c
ls gpurun_out/r5n/dump 2>/dev/null | head
