#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2k
mkdir -p $OUT
for i in 1 2; do echo "== c2"; bash tools/gpu_quick.sh r2k/c2_$i --steps 20 --warmup 3; done
echo "== c3_dog"; bash tools/gpu_quick.sh r2k/c3_dog --config c3_dog --steps 20 --warmup 3
echo "== c3_paul"; bash tools/gpu_quick.sh r2k/c3_paul --config c3_paul --steps 20 --warmup 3
for sh in 0/2 0/4 0/8 3/8; do echo "== shard $sh"; bash tools/gpu_quick.sh r2k/shard_${sh/\//_} --shard $sh --force-dist --steps 40 --warmup 5; done
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save or every_row" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_ols.log
