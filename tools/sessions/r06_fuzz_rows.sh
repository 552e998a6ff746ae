# round 6: long fuzz of the rows whose kernels changed this round (tiled projections uint8 / float32 / uint8-out, normalize's stretch pass)
cd /root/repo
for seed in 601 602 603 604 605 606; do FUZZ_KINDS=project,normalize,project timeout 600 python tools/fuzz_rows.py $seed 200 2>&1 | tail -1; done
for seed in 611 612; do timeout 600 python tools/fuzz_rows.py $seed 150 2>&1 | grep -v "^ok" | tail -3; done
