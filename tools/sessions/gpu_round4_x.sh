#!/bin/bash
# round 4, session x: randomised differential runs on WIDE window grids (more columns than the walking kernels' job strips)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4x
export FUZZ_WIDE=1
for s in 701 702 703; do timeout 400 python tools/fuzz_parity.py $s 100 > gpurun_out/r4x/parity_$s.log 2>&1; echo "parity $s: $(tail -1 gpurun_out/r4x/parity_$s.log | cut -c1-200)"; grep FAIL gpurun_out/r4x/parity_$s.log | head -3 | cut -c1-300; done
for m in ensemble timestep planes; do for s in 711 712 713; do FUZZ_MODE=$m FUZZ_DUMP=$R/gpurun_out/r4x/dump timeout 400 python tools/fuzz_modes.py $s 60 > gpurun_out/r4x/${m}_$s.log 2>&1; echo "$m $s: $(grep -E 'cases,' gpurun_out/r4x/${m}_$s.log | tail -1 | cut -c1-200)"; grep FAIL gpurun_out/r4x/${m}_$s.log | head -3 | cut -c1-400; done; done
