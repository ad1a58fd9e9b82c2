#!/bin/bash
# round 3, session c: prologue ablation (no spectrum / table loads in the input stage), 4096-point overlap-save tiles
export TMPDIR=/tmp
OUT=gpurun_out/r3c
mkdir -p $OUT
echo "== c2 default"; bash tools/gpu_quick.sh r3c/c2 --steps 30 --warmup 3
bash tools/gpu_variants.sh r3c/var "--steps 30 --warmup 3" abl_pro
echo "== c2 ols_tile 4096"; bash tools/gpu_quick.sh r3c/c2_t4096 --steps 30 --warmup 3 --opt ols_tile=4096
echo "== ols sweep 8192"; timeout 300 python tools/ols_sweep.py > $OUT/ols_sweep_8192.txt 2>&1; cat $OUT/ols_sweep_8192.txt
echo "== ols sweep 4096"; timeout 300 python tools/ols_sweep.py --opt ols_tile=4096 > $OUT/ols_sweep_4096.txt 2>&1; cat $OUT/ols_sweep_4096.txt
