#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2c
mkdir -p $OUT
python tools/ols_sweep.py --prec 64 > $OUT/ols_sweep_fp64.txt 2>&1; cat $OUT/ols_sweep_fp64.txt
python tools/ols_sweep.py --prec 32 --mother 2 > $OUT/ols_sweep_fp32_dog.txt 2>&1; cat $OUT/ols_sweep_fp32_dog.txt
for cfg in c2 c3_paul c3_dog; do
  echo "== $cfg"; bash tools/gpu_quick.sh r2c/${cfg} --config $cfg --steps 20 --warmup 3
done
echo "== c2 ols_side=0"; bash tools/gpu_quick.sh r2c/c2_side0 --opt ols_side=0 --steps 20 --warmup 3
echo "== c2 overlap_narrow=0"; bash tools/gpu_quick.sh r2c/c2_on0 --opt overlap_narrow=0 --steps 20 --warmup 3
echo "== c2 ols=0"; bash tools/gpu_quick.sh r2c/c2_ols0 --opt ols=0 --steps 20 --warmup 3
echo "== c2 again"; bash tools/gpu_quick.sh r2c/c2_b --steps 20 --warmup 3
