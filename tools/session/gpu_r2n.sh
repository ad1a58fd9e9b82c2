#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2fs
mkdir -p $OUT
echo "== c2 1 GPU"; bash tools/gpu_quick.sh r2fs/c2 --steps 20 --warmup 3 | cut -c1-60
for G in 2 4 8; do
  for ((r=0; r<G; r++)); do echo "== balanced shard $r/$G"; bash tools/gpu_quick.sh r2fs/b_${r}_$G --shard $r/$G --force-dist --steps 40 --warmup 5 | cut -c1-400; done
done
for n in 16 17 18; do echo "== logn $n"; bash tools/gpu_quick.sh r2fs/n$n --logn $n --steps 20 --warmup 3 | cut -c1-60; done
