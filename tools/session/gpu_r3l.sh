#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3l
mkdir -p $OUT
for sh in 2/8 5/8 0/8 4/8; do
  for p in 1 2 3; do echo "== shard $sh pipeline $p"; bash tools/gpu_quick.sh r3l/p${p}_${sh/\//_} --shard $sh --force-dist --steps 60 --warmup 6 --pipeline $p | cut -c1-60; done
done
