#!/bin/bash
# round 3, session d: overlap-save tile sizes 1024 / 2048 (sweep per scale), band-limited K <= 512 rows on 4096-point tiles in fp64
export TMPDIR=/tmp
OUT=gpurun_out/r3d
mkdir -p $OUT
echo "== c2 default"; bash tools/gpu_quick.sh r3d/c2 --steps 30 --warmup 3
echo "== c2 narrow_small=2"; bash tools/gpu_quick.sh r3d/c2_ns2 --steps 30 --warmup 3 --opt narrow_small=2
for t in 2048 1024; do echo "== ols sweep $t"; timeout 300 python tools/ols_sweep.py --opt ols_tile=$t > $OUT/ols_sweep_$t.txt 2>&1; cat $OUT/ols_sweep_$t.txt; done
for t in 8192 4096 2048; do echo "== ols sweep fp32 dog $t"; timeout 300 python tools/ols_sweep.py --prec 32 --mother 2 --opt ols_tile=$t > $OUT/ols_sweep_f32dog_$t.txt 2>&1; cat $OUT/ols_sweep_f32dog_$t.txt; done
