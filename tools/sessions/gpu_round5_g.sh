#!/bin/bash
# round 5, session g: XCD order by windows (LSPIV_XCD_ORDER=1) against the contiguous-range order (0): same bits? and the anchor lengths
# 25 / 125 under both orders at 300 / 1000 / 1040 pairs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for o in 0 1; do
  export LSPIV_XCD_ORDER=$o
  python tools/piv_hash.py 32 16 130 | sed "s/^/order $o /"; python tools/piv_hash.py 64 48 130 | sed "s/^/order $o /"; python tools/piv_hash.py 32 16 55 270 325 | sed "s/^/order $o /"; python tools/piv_hash.py 24 12 60 500 700 | sed "s/^/order $o /"
  python tools/ens_hash.py 64 48 130 | sed "s/^/order $o /"; python tools/ens_hash.py 32 16 130 | sed "s/^/order $o /"
done
LSPIV_XCD_ORDER=1 timeout 900 python -m pytest tests/test_gpu_strip_order.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x 2>&1 | tail -3
for P in 300 1000 1040; do
  for w in 25 125; do
    for o in 0 1; do
      export LSPIV_XCD_ORDER=$o
      LSPIV_WALK=$w python tools/ens_launch.py 64 48 $P 4 | cut -c88-140 | sed "s/^/ens64 P=$P anchor $w order $o: /"
      LSPIV_WALK=$w python tools/ens_launch.py 32 16 $P 6 | cut -c88-140 | sed "s/^/ens32 P=$P anchor $w order $o: /"
      LSPIV_WALK=$w LSPIV_RESCUE=0 python tools/ab_time.py --window 64 --overlap 48 --pairs $P --tag "c3 P=$P anchor $w order $o" | tail -1
      LSPIV_WALK=$w LSPIV_RESCUE=0 python tools/ab_time.py --window 32 --overlap 16 --pairs $P --tag "c2 P=$P anchor $w order $o" | tail -1
    done
  done
done
