#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3i
mkdir -p $OUT
for c in c3_dog c3_paul; do echo "== $c product"; bash tools/gpu_quick.sh r3i/$c --config $c --steps 30 --warmup 3; done
bash tools/gpu_variants.sh r3i/dog "--steps 30 --warmup 3 --config c3_dog" f32h4 f32h5 f32h8 f32n5 f32n6
bash tools/gpu_variants.sh r3i/paul "--steps 30 --warmup 3 --config c3_paul" f32h4 f32h5 f32h8 f32n5 f32n6
echo "== c2 product"; bash tools/gpu_quick.sh r3i/c2 --steps 30 --warmup 3
bash tools/gpu_variants.sh r3i/c2v "--steps 30 --warmup 3" f64h5
for w in 50 150 200 300; do echo "== c2 ols_fwd_weight=$w"; bash tools/gpu_quick.sh r3i/c2_w$w --steps 30 --warmup 3 --opt ols_fwd_weight=$w; done
for w in 150 250; do echo "== dog ols_fwd_weight=$w"; bash tools/gpu_quick.sh r3i/dog_w$w --config c3_dog --steps 30 --warmup 3 --opt ols_fwd_weight=$w; done
