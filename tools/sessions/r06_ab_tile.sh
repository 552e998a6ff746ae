cd /root/repo
run() { LSPIV_LIBRARY=$1 timeout 120 python tools/rows_launch.py project 30 201 | sed -E 's/^project: ([0-9.]+) ms.*= ([0-9.]+) %.*/\1 ms \2 %/'; }
for round in 1 2 3; do
  for v in base f4 f16 nt w1 w2 w8 f4nt; do
    if [ $v = base ]; then lib=""; else lib=build/ab/lib_t_$v.so; fi
    echo "$round $v $(run $lib)"
  done
  echo "$round base_r2 $(LSPIV_PROJECT_TILE_RMAX=2 run '')"
done
