#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3t
mkdir -p $OUT
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3t/$tag --steps 100 --warmup 5 "$@" | sed -E 's/dom=.*kernels=/k=/; s/split=.*//' | cut -c1-400; }
q c2_a; q c2_b; q dog --config c3_dog; q paul --config c3_paul
