#!/bin/bash
export TMPDIR=/tmp
echo "== c3_dog base (LB 6)"; bash tools/gpu_quick.sh r2t/dog --config c3_dog --steps 20 --warmup 3 | cut -c1-300
bash tools/gpu_variants.sh r2t "--config c3_dog --steps 20 --warmup 3" olsf32_8 | cut -c1-300
echo "== c3_dog base (LB 6)"; bash tools/gpu_quick.sh r2t/dog2 --config c3_dog --steps 20 --warmup 3 | cut -c1-300
