#!/bin/bash
# round 3, session j: every rank's share of config 2 at G = 2, 4, 8 on one GPU (cost-balanced contiguous shards)
export TMPDIR=/tmp
OUT=gpurun_out/r3j
mkdir -p $OUT
echo "== c2 1 GPU"; bash tools/gpu_quick.sh r3j/c2 --steps 30 --warmup 3
for G in 2 4 8; do
  for ((r=0; r<G; r++)); do echo "== balanced shard $r/$G"; bash tools/gpu_quick.sh r3j/b_${r}_$G --shard $r/$G --force-dist --steps 40 --warmup 5 | cut -c1-520; done
done
for sh in 0/8 4/8; do echo "== interleaved shard $sh"; bash tools/gpu_quick.sh r3j/i_${sh/\//_} --shard $sh --force-dist --partition interleaved --steps 40 --warmup 5 | cut -c1-400; done
