#!/bin/bash
# K = 2048 rows on 8192-point tiles (narrow_big = 2: two workgroups per CU, 64-byte store segments) against the 16384-point
# tiles (1: one workgroup per CU) and multi-term K = 1024 (0 with narrow_terms = 8); interleaved repeats
export TMPDIR=/tmp
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3at/$tag --steps 200 --warmup 5 "$@" | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330; }
for i in 1 2 3; do
  q c2_big1_$i
  q c2_big2_$i --opt narrow_big=2
done
q c2_big2_bt8 --opt narrow_big=2 --opt big_terms=8
q c2_big1_bt8 --opt big_terms=8
