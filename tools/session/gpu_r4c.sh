#!/bin/bash
# round 4, session c: band-limited rows in polynomial form (poly) -- parity and A/B
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 300 python tools/session/r4a_check.py 256 > $OUT/check.txt 2>&1; grep -v "row" $OUT/check.txt | tail -12
bash tools/gpu_quick.sh r4c/c2
bash tools/gpu_quick.sh r4c/c2_nopoly --opt poly=0
for d in 4 6 10 12; do bash tools/gpu_quick.sh r4c/c2_deg$d --opt poly_degree=$d; done
for c in c3_paul c3_dog; do
  bash tools/gpu_quick.sh r4c/${c} --config $c
  bash tools/gpu_quick.sh r4c/${c}_nopoly --config $c --opt poly=0
  bash tools/gpu_quick.sh r4c/${c}_deg4 --config $c --opt poly_degree=4
done
bash tools/gpu_quick.sh r4c/c2_t16 --opt tolerance_neglog10=16
