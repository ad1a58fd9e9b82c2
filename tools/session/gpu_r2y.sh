#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
echo "== dog base"; bash tools/gpu_quick.sh r2y/dog_$i --config c3_dog --steps 30 --warmup 3 | cut -c1-200
bash tools/gpu_variants.sh r2y "--config c3_dog --steps 30 --warmup 3" nh6 nh5 | cut -c1-200
done
echo "== paul base"; bash tools/gpu_quick.sh r2y/paul --config c3_paul --steps 30 --warmup 3 | cut -c1-200
bash tools/gpu_variants.sh r2y "--config c3_paul --steps 30 --warmup 3" nh6 | cut -c1-200
