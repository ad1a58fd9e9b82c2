#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2d
mkdir -p $OUT
for i in 1 2; do
echo "== c2 default"; bash tools/gpu_quick.sh r2d/c2_$i --steps 20 --warmup 3
echo "== c2 ols_early=0"; bash tools/gpu_quick.sh r2d/c2_early0_$i --opt ols_early=0 --steps 20 --warmup 3
done
echo "== c3_dog"; bash tools/gpu_quick.sh r2d/c3_dog --config c3_dog --steps 20 --warmup 3
echo "== c3_dog early0"; bash tools/gpu_quick.sh r2d/c3_dog_e0 --config c3_dog --steps 20 --warmup 3 --opt ols_early=0
echo "== c3_paul"; bash tools/gpu_quick.sh r2d/c3_paul --config c3_paul --steps 20 --warmup 3
cp pycwt_amd/libcwt_hip.so /tmp/keep.so
for v in abl1 abl2; do
  cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  echo "== variant $v"; python tools/ols_sweep.py --prec 64 2>&1 | grep ols/
done
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
