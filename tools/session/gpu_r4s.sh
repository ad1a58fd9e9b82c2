#!/bin/bash
export TMPDIR=/tmp
for v in "" "--opt ols_hold=0" "--opt ols_hold=0 --opt ols_small_max_halo=1024" "--opt ols_hold=0 --opt ols_small_max_halo=768" "" "--opt ols_hold=0 --opt ols_small_max_halo=1024"; do
  bash tools/gpu_quick.sh r4s/c2_$(echo $v | tr -d ' =-') --no-live-traffic $v | cut -c1-60
done
for c in c3_dog c3_paul; do for v in "" "--opt ols_hold=0" "--opt ols_hold=0 --opt ols_small_max_halo=1024"; do
  bash tools/gpu_quick.sh r4s/${c}_$(echo $v | tr -d ' =-') --config $c --no-live-traffic $v | cut -c1-60
done; done
