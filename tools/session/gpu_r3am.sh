#!/bin/bash
# fp32 option sweeps at the new launch bound: half-tile halo threshold, alternating launch order, four-tile blocks
export TMPDIR=/tmp
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3am/$tag --steps 200 --warmup 5 "$@" | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330; }
for c in c3_dog c3_paul; do
  q ${c}_base --config $c
  q ${c}_mix --config $c --opt narrow_mix=1
  q ${c}_h768 --config $c --opt ols_small_max_halo=768
  q ${c}_h1024 --config $c --opt ols_small_max_halo=1024
  q ${c}_h256 --config $c --opt ols_small_max_halo=256
  q ${c}_big2 --config $c --opt ols_big=2
  q ${c}_big2_5k --config $c --opt ols_big=2 --opt ols_big4_max_halo=5120
  q ${c}_base2 --config $c
done
