#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do echo "== c3_dog base"; bash tools/gpu_quick.sh r2s/dog_$i --config c3_dog --steps 20 --warmup 3 | cut -c1-300; done
bash tools/gpu_variants.sh r2s "--config c3_dog --steps 20 --warmup 3" olsf32_6 olsf32_5 | cut -c1-300
bash tools/gpu_variants.sh r2s "--config c3_paul --steps 20 --warmup 3" olsf32_6 | cut -c1-300
echo "== c3_paul base"; bash tools/gpu_quick.sh r2s/paul --config c3_paul --steps 20 --warmup 3 | cut -c1-300
