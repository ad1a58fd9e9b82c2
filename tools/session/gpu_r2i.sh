#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2i
mkdir -p $OUT
for i in 1 2; do echo "== c2"; bash tools/gpu_quick.sh r2i/c2_$i --steps 20 --warmup 3; done
echo "== c2 ols_big=0"; bash tools/gpu_quick.sh r2i/c2_big0 --opt ols_big=0 --steps 20 --warmup 3
echo "== c2 bmh 1024"; bash tools/gpu_quick.sh r2i/c2_bmh1024 --opt ols_big_min_halo=1024 --steps 20 --warmup 3
echo "== c2 bmh 512"; bash tools/gpu_quick.sh r2i/c2_bmh512 --opt ols_big_min_halo=512 --steps 20 --warmup 3
echo "== c3_dog"; bash tools/gpu_quick.sh r2i/c3_dog --config c3_dog --steps 20 --warmup 3
echo "== c3_dog bmh 512"; bash tools/gpu_quick.sh r2i/c3_dog_bmh512 --config c3_dog --steps 20 --warmup 3 --opt ols_big_min_halo=512
echo "== c3_paul"; bash tools/gpu_quick.sh r2i/c3_paul --config c3_paul --steps 20 --warmup 3
for sh in 0/2 0/4 0/8 3/8; do echo "== shard $sh"; bash tools/gpu_quick.sh r2i/shard_${sh/\//_} --shard $sh --force-dist --steps 20 --warmup 3; done
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_ols.log
