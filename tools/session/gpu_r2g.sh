#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2g
mkdir -p $OUT
for cfg in c3_paul c3_dog; do
for i in 1 2; do
echo "== $cfg default (tile 8192)"; bash tools/gpu_quick.sh r2g/${cfg}_$i --config $cfg --steps 20 --warmup 3
echo "== $cfg tile 16384"; bash tools/gpu_quick.sh r2g/${cfg}_t16_$i --config $cfg --opt ols_tile=16384 --steps 20 --warmup 3
done
echo "== $cfg big0"; bash tools/gpu_quick.sh r2g/${cfg}_big0 --config $cfg --opt ols_big=0 --steps 20 --warmup 3
echo "== $cfg bmh 512"; bash tools/gpu_quick.sh r2g/${cfg}_bmh512 --config $cfg --opt ols_big_min_halo=512 --steps 20 --warmup 3
done
python tools/ols_sweep.py --prec 32 --mother 2 > $OUT/ols_sweep_fp32_dog.txt 2>&1; cat $OUT/ols_sweep_fp32_dog.txt
python tools/ols_sweep.py --prec 32 --mother 1 > $OUT/ols_sweep_fp32_paul.txt 2>&1; cat $OUT/ols_sweep_fp32_paul.txt
