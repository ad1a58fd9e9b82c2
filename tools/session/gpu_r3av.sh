#!/bin/bash
# fp32: the K = 1024 band-limited rows (1024-thread tiles) on the second side stream beside the K <= 512 rows instead of
# behind them (narrow_split = 1); interleaved repeats
export TMPDIR=/tmp
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3av/$tag --steps 200 --warmup 5 "$@" | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-60; }
for i in 1 2 3 4; do
  for c in c3_dog c3_paul; do
    q ${c}_base_$i --config $c
    q ${c}_split_$i --config $c --opt narrow_split=1
  done
done
