#!/bin/bash
# round 6, last kernels: fuzz of the four-column blur and of project_cv in one kernel (widths that are multiples of four), then all kinds
cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/fuzz_blur_cv; mkdir -p $OUT
for seed in 621 622 623 624; do FUZZ_W4=1 FUZZ_KINDS=blur,project_cv timeout 900 python tools/fuzz_rows.py $seed 200 2>&1 | grep -v "^ok" | tail -4; done | tee $OUT/log.txt
for seed in 631 632; do FUZZ_W4=1 FUZZ_DIST=0.3 FUZZ_KINDS=project_cv timeout 900 python tools/fuzz_rows.py $seed 150 2>&1 | grep -v "^ok" | tail -4; done | tee -a $OUT/log.txt
for seed in 641 642; do timeout 900 python tools/fuzz_rows.py $seed 150 2>&1 | grep -v "^ok" | tail -3; done | tee -a $OUT/log.txt
# after remap_fused_f32_kernel: project_cv again (uint8 and float32 frames through the one-kernel path)
for seed in 651 652 653; do FUZZ_W4=1 FUZZ_KINDS=project_cv timeout 600 python tools/fuzz_rows.py $seed 200 2>&1 | grep -v "^ok" | tail -3; done | tee -a $OUT/log.txt
FUZZ_W4=1 FUZZ_DIST=0.3 FUZZ_KINDS=project_cv timeout 600 python tools/fuzz_rows.py 654 200 2>&1 | grep -v "^ok" | tail -3 | tee -a $OUT/log.txt
# after blur_stripr_kernel: the filters over every radius class (windows up to 15)
for seed in 661 662 663 664; do FUZZ_KINDS=blur timeout 600 python tools/fuzz_rows.py $seed 200 2>&1 | grep -v "^ok" | tail -3; done | tee -a $OUT/log.txt
for seed in 671 672 673; do FUZZ_BIGWDW=1 FUZZ_KINDS=blur timeout 600 python tools/fuzz_rows.py $seed 200 2>&1 | grep -v "^ok" | tail -3; done | tee -a $OUT/log.txt
