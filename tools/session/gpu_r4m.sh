#!/bin/bash
export TMPDIR=/tmp
for v in "" "--opt poly_max_logk=13" "--opt poly_max_logk=12" "--opt poly_max_logk=13 --opt poly_degree=12" "--opt poly_max_logk=13 --opt ols_big=2"; do
  bash tools/gpu_quick.sh r4m/c2_$(echo $v | tr -d ' =-') --no-live-traffic $v
done
for v in "" "--opt poly_max_logk=12"; do
  bash tools/gpu_quick.sh r4m/paul_$(echo $v | tr -d ' =-') --config c3_paul --no-live-traffic $v
  bash tools/gpu_quick.sh r4m/dog_$(echo $v | tr -d ' =-') --config c3_dog --no-live-traffic $v
done
