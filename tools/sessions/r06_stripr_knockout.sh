cd $GRAFT_REPO_ROOT
for lib in tree SR_SKIP_LDS SR_SKIP_STORES SR_SKIP_BOTH; do
  if [ $lib = tree ]; then python tools/filters_bench.py 201; else LSPIV_LIBRARY=build/ab/lib_$lib.so python tools/filters_bench.py 201; fi 2>&1 | grep -E "edge 5\|9|k=11|edge 13" | cut -c1-60 | sed "s/^/$lib /"
done
