#!/bin/bash
# round 4, session e: polynomial rows, second form (planes + LDS-staged coefficient sets, two passes per workgroup)
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 300 python tools/session/r4a_check.py 256 > $OUT/check.txt 2>&1; grep -v "row" $OUT/check.txt | tail -12
bash tools/gpu_quick.sh r4e/c2
bash tools/gpu_quick.sh r4e/c2_nopoly --opt poly=0
for d in 6 10 12; do bash tools/gpu_quick.sh r4e/c2_deg$d --opt poly_degree=$d; done
for c in c3_paul c3_dog; do
  bash tools/gpu_quick.sh r4e/${c} --config $c
  bash tools/gpu_quick.sh r4e/${c}_deg4 --config $c --opt poly_degree=4
  bash tools/gpu_quick.sh r4e/${c}_deg12 --config $c --opt poly_degree=12
done
bash tools/gpu_quick.sh r4e/c2_t16 --opt tolerance_neglog10=16
