#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3k
mkdir -p $OUT
S="--opt ols_early=0 --opt ols_side=0 --opt overlap_narrow=0"
for sh in 2/8 5/8 0/8 4/8 1/4; do
  echo "== shard $sh default"; bash tools/gpu_quick.sh r3k/d_${sh/\//_} --shard $sh --force-dist --steps 40 --warmup 5 | cut -c1-60
  echo "== shard $sh serialized"; bash tools/gpu_quick.sh r3k/s_${sh/\//_} --shard $sh --force-dist --steps 40 --warmup 5 $S | cut -c1-60
  echo "== shard $sh serialized, no dist"; bash tools/gpu_quick.sh r3k/n_${sh/\//_} --shard $sh --steps 40 --warmup 5 $S | cut -c1-60
done
for i in 1 2; do echo "== c2 default"; bash tools/gpu_quick.sh r3k/c2d$i --steps 30 --warmup 3 | cut -c1-60; echo "== c2 serialized"; bash tools/gpu_quick.sh r3k/c2s$i --steps 30 --warmup 3 $S | cut -c1-60; done
for c in c3_dog c3_paul; do echo "== $c default"; bash tools/gpu_quick.sh r3k/${c}d --config $c --steps 30 --warmup 3 | cut -c1-60; echo "== $c serialized"; bash tools/gpu_quick.sh r3k/${c}s --config $c --steps 30 --warmup 3 $S | cut -c1-60; done
