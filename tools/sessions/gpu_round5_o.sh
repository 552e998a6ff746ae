#!/bin/bash
# round 5, session o: the last source changes (masks on single-cell grids, Ensemble.accumulate(out=), bench tests) through the GPU suite + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 120 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -4
for s in 1201 1202; do FUZZ_MODE=ensemble timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1; done
for s in 1211 1212; do timeout 200 python tools/fuzz_modes.py $s 80 2>&1 | grep -E "FAIL|cases," | tail -1; done
