#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do
echo "== new c2"; bash tools/gpu_quick.sh r2v/c2_$i --steps 30 --warmup 3 | cut -c1-330
bash tools/gpu_variants.sh r2v "--steps 30 --warmup 3" base | cut -c1-330
done
echo "== new dog"; bash tools/gpu_quick.sh r2v/dog --config c3_dog --steps 30 --warmup 3 | cut -c1-330
bash tools/gpu_variants.sh r2v "--config c3_dog --steps 30 --warmup 3" base | cut -c1-330
