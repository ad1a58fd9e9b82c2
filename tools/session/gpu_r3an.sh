#!/bin/bash
# fp32: alternating launch order (narrow_mix) and half-tile halo threshold 256 / 384, three interleaved repeats
export TMPDIR=/tmp
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3an/$tag --steps 200 --warmup 5 "$@" | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-60; }
for i in 1 2 3; do
for c in c3_dog c3_paul; do
  q ${c}_base_$i --config $c
  q ${c}_mix_$i --config $c --opt narrow_mix=1
  q ${c}_mix_h256_$i --config $c --opt narrow_mix=1 --opt ols_small_max_halo=256
  q ${c}_mix_h384_$i --config $c --opt narrow_mix=1 --opt ols_small_max_halo=384
done
done
