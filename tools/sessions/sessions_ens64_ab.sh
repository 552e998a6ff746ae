# A/B of the 64 x 64 ensemble kernel's slot layout (round 6): the tree (hot halves in an array of their own) against
# build/ab/lib_ens64_interleaved.so (-DLSPIV_ENS_SPLIT_HALVES=0): correctness, time, HBM bytes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ens64_ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_strip_order.py tests/test_gpu_fullsize.py tests/test_gpu_shard.py -m gpu -q -k "ensemble" 2>&1 | tail -3
for i in 1 2; do
  python tools/ens_launch.py 64 48 1000 6 | cut -c1-200
  LSPIV_LIBRARY=build/ab/lib_ens64_interleaved.so python tools/ens_launch.py 64 48 1000 6 | cut -c1-200
done
bash tools/profile_ens.sh r06split 64 48 > gpurun_out/ens64_ab/split.log 2>&1
LSPIV_LIBRARY=$R/build/ab/lib_ens64_interleaved.so bash tools/profile_ens.sh r06inter 64 48 > gpurun_out/ens64_ab/inter.log 2>&1
for t in r06split r06inter; do python3 - gpurun_out/prof_${t}_ens64 <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/pmc_*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "walk_ensemble" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v)/len(v)*1024/1e9,3) for k,v in acc.items() if k in ("FETCH_SIZE","WRITE_SIZE")}, {k: round(sum(v)/len(v)/1e6,1) for k,v in acc.items() if k.startswith("TCC")})
PY
grep walk_ensemble gpurun_out/prof_${t}_ens64/trace_kernel_stats.csv | cut -d, -f1-5 | cut -c60-
done
