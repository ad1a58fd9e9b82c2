#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do
echo "== c2 default"; bash tools/gpu_quick.sh r2w/c2_$i --steps 30 --warmup 3 | cut -c1-330
echo "== c2 ols_big only beyond the plain reach"; bash tools/gpu_quick.sh r2w/c2_big_$i --steps 30 --warmup 3 --opt ols_big=1 --opt ols_big_min_halo=2112 | cut -c1-330
done
