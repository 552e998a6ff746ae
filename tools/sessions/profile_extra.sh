#!/bin/bash
# kernel-trace stats for the other kernels (64x64 FFT, projection, filters, masks, ensemble, the other window sizes) -> gpurun_out/prof_extra_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1}
OUT=$R/gpurun_out/prof_extra_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$name -o $name -- "$@" > $OUT/$name.log 2>&1; \
        find /tmp/px_$name -name "*kernel_stats.csv" -exec cp {} $OUT/ \; ; }
run fft64 python $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --window 64 --overlap 48 --pairs 300
run project python $R/tools/project_bench.py 201
run filters python $R/tools/filters_bench.py 201
run dtypes python $R/tools/dtype_bench.py 100
run masks python $R/tools/mask_bench.py 1000
run ensemble python $R/tools/ensemble_bench.py 301
run other_windows python $R/tools/direct_bench.py 200
run win24_1080p python $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --window 24 --overlap 12 --pairs 500
for f in $OUT/*kernel_stats.csv; do echo == $f; grep -v "synth_" $f | cut -c1-170; done
