#!/bin/bash
# round 4, session o: 32 x 32 ensemble kernel without the scheduling barriers between its phases (it has registers to spare at two
# waves per SIMD): build/ab/lib_ens32relax.so (launch bounds 2) and lib_ens32relax3.so (bounds 3) against the default, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lib in relax relax3; do LSPIV_LIBRARY=$R/build/ab/lib_ens32$lib.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "ensemble" --timeout 300 2>&1 | tail -1; done
for round in 1 2 3; do
  python tools/ens_launch.py 32 16 1000 8 | tail -1 | cut -c1-140
  LSPIV_LIBRARY=$R/build/ab/lib_ens32relax.so python tools/ens_launch.py 32 16 1000 8 | tail -1 | cut -c1-140 | sed 's/^/ [relax]  /'
  LSPIV_LIBRARY=$R/build/ab/lib_ens32relax3.so python tools/ens_launch.py 32 16 1000 8 | tail -1 | cut -c1-140 | sed 's/^/ [relax3] /'
done
