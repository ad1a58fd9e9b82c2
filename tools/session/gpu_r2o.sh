#!/bin/bash
export TMPDIR=/tmp
for n in 16 17; do
echo "== logn $n default"; bash tools/gpu_quick.sh r2o/n$n --logn $n --steps 50 --warmup 5 | cut -c1-400
echo "== logn $n overlap_narrow=0"; bash tools/gpu_quick.sh r2o/n${n}_on0 --logn $n --steps 50 --warmup 5 --opt overlap_narrow=0 | cut -c1-100
echo "== logn $n ols_min_logn=15"; bash tools/gpu_quick.sh r2o/n${n}_ols --logn $n --steps 50 --warmup 5 --opt ols_min_logn=15 | cut -c1-100
done
echo "== c2"; bash tools/gpu_quick.sh r2o/c2 --steps 20 --warmup 3 | cut -c1-100
