#!/bin/bash
# forward FFT of the one signal on half-size tiles (fwd_small = 1: 256 / 128 workgroups instead of 128 / 64); interleaved
export TMPDIR=/tmp
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3au/$tag --steps 200 --warmup 5 "$@" | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-150; }
for i in 1 2 3; do
  for c in c2 c3_dog c3_paul; do
    q ${c}_small_$i --config $c
    q ${c}_full_$i --config $c --opt fwd_small=0
  done
done
