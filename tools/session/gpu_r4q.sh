#!/bin/bash
export TMPDIR=/tmp
for v in "" "--opt ols_small_max_halo=768" "--opt ols_small_max_halo=1024" "" "--opt ols_small_max_halo=1024"; do
  bash tools/gpu_quick.sh r4q/c2_$(echo $v | tr -d ' =-') --no-live-traffic $v | cut -c1-330
done
for v in "" "--opt ols_small_max_halo=1024"; do
  bash tools/gpu_quick.sh r4q/dog_$(echo $v | tr -d ' =-') --config c3_dog --no-live-traffic $v | cut -c1-330
  bash tools/gpu_quick.sh r4q/paul_$(echo $v | tr -d ' =-') --config c3_paul --no-live-traffic $v | cut -c1-330
done
