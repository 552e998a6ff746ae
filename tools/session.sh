#!/bin/bash
# One GPU session (run through gpurun): tools/session.sh <name> -- the steps of a named session, output under gpurun_out/<name>/.
# Replaces the per-session gpu_round*_*.sh scripts of rounds 2-5 (archived under tools/sessions/).
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-tests}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd $R
case $NAME in
  tests)        # the GPU suite as the driver runs it, then the default bench line
    timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; head -c 600 $OUT/bench.json;;
  new-tests)    # only this round's GPU tests (fast iteration)
    timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -x -q 2>&1 | tail -30 | tee $OUT/pytest.log;;
  bench)
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err; head -c 400 $OUT/bench.json;;
  rows)         # the HBM-bound rows: timings, then rocprofv3 passes per row (tools/profile.sh with another kernel filter)
    for row in project project_nn project_u8 project_f32 project_cv project_cv_f32 time_diff normalize smooth edge_detect edge_detect_6_10 smooth_f32 edge_detect_f32; do
      python tools/rows_launch.py $row 30 201 | tee -a $OUT/rows.log
    done
    for row in ${ROWS:-project project_nn project_u8 project_f32 project_cv project_cv_f32 time_diff normalize smooth edge_detect edge_detect_6_10 smooth_f32 edge_detect_f32}; do
      case $row in project|project_nn|project_u8|project_f32) re='project_';; project_cv|project_cv_f32) re='remap_';; time_diff) re='time_diff_';; normalize) re='norm|sample_mean|frame_minmax';; smooth|edge_detect|edge_detect_6_10|smooth_f32|edge_detect_f32) re='blur_';; esac
      PROFILE_KF="--kernel-include-regex $re" PROFILE_CMD="python $R/tools/rows_launch.py $row 30 201" bash tools/profile.sh r06_rows_$row > $OUT/profile_$row.log 2>&1
      SUMMARY_KERNELS="$re" SUMMARY_ROW=$row SUMMARY_ROW_FRAMES=201 SUMMARY_CALLS=32 python tools/summarize_profile.py gpurun_out/prof_r06_rows_$row r06_rows_$row > $OUT/summary_$row.log 2>&1
      cp profiles/r06_rows_${row}_summary.json $OUT/ 2>/dev/null
      for f in gpurun_out/prof_r06_rows_$row/trace_kernel_stats.csv; do cp $f $OUT/r06_rows_${row}_trace_kernel_stats.csv 2>/dev/null; done
    done;;
  profiles)     # the whole committed profile set of a round: PIV configs + ensemble kernels (tools/profile_all.sh), then the rows
    bash tools/profile_all.sh ${TAG:-r06} > $OUT/profile_all.log 2>&1; tail -25 $OUT/profile_all.log
    mkdir -p $OUT/profiles; cp profiles/${TAG:-r06}_* $OUT/profiles/ 2>/dev/null
    for d in gpurun_out/prof_${TAG:-r06}_*; do t=$(basename $d | sed 's/^prof_//'); for f in $d/*counter_collection.csv $d/trace_kernel_stats.csv; do [ -f $f ] && cp $f $OUT/profiles/${t}_$(basename $f | sed 's/_counter_collection/_counter_collection/'); done; done
    NAME=rows bash tools/session.sh rows > $OUT/rows.log 2>&1; tail -8 gpurun_out/rows/rows.log;;
  *) echo "unknown session $NAME"; exit 2;;
esac
