#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2l
mkdir -p $OUT
echo "== c2 1 GPU"; bash tools/gpu_quick.sh r2l/c2 --steps 20 --warmup 3
for G in 2 4 8; do
  for ((r=0; r<G; r++)); do echo "== balanced shard $r/$G"; bash tools/gpu_quick.sh r2l/b_${r}_$G --shard $r/$G --force-dist --steps 40 --warmup 5 | cut -c1-420; done
done
for sh in 0/8 4/8 0/4; do echo "== interleaved shard $sh"; bash tools/gpu_quick.sh r2l/i_${sh/\//_} --shard $sh --force-dist --partition interleaved --steps 40 --warmup 5 | cut -c1-300; done
