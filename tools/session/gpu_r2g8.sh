#!/bin/bash
export TMPDIR=/tmp
echo "== c2 1 GPU"; bash tools/gpu_quick.sh r2g8/c2 --steps 20 --warmup 3 | cut -c1-60
for ((r=0; r<8; r++)); do echo "== balanced shard $r/8"; bash tools/gpu_quick.sh r2g8/b_${r}_8 --shard $r/8 --force-dist --steps 40 --warmup 5 | cut -c1-60; done
