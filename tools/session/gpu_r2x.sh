#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do
echo "== c2 default"; bash tools/gpu_quick.sh r2x/c2_$i --steps 30 --warmup 3 | cut -c1-60
echo "== c2 chain first, then narrow || ols"; bash tools/gpu_quick.sh r2x/c2_cf_$i --steps 30 --warmup 3 --opt chain_first=1 --opt ols_early=0 --opt ols_side=0 | cut -c1-60
echo "== c2 chain first, ols early"; bash tools/gpu_quick.sh r2x/c2_cf2_$i --steps 30 --warmup 3 --opt chain_first=1 | cut -c1-60
done
