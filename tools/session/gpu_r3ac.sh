#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3ac
echo "== c2 1 GPU"; bash tools/gpu_quick.sh r3ac/c2 --steps 100 --warmup 5
for G in 8 4; do for ((r=0; r<G; r++)); do echo "== balanced shard $r/$G"; bash tools/gpu_quick.sh r3ac/b_${r}_$G --shard $r/$G --force-dist --steps 60 --warmup 5 | cut -c1-520; done; done
