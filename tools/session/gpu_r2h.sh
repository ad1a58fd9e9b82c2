#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2h
mkdir -p $OUT
for c in 0 17 6 4; do
echo "== c2 chunk_rows=$c"; bash tools/gpu_quick.sh r2h/c2_ch$c --opt chunk_rows=$c --steps 20 --warmup 3
done
echo "== c2 overlap=1 chunk 6"; bash tools/gpu_quick.sh r2h/c2_ov6 --opt overlap=1 --opt chunk_rows=6 --steps 20 --warmup 3
echo "== c2 fwd_weight 200"; bash tools/gpu_quick.sh r2h/c2_w200 --opt ols_fwd_weight=200 --steps 20 --warmup 3
echo "== c2 fwd_weight 300"; bash tools/gpu_quick.sh r2h/c2_w300 --opt ols_fwd_weight=300 --steps 20 --warmup 3
echo "== c3_dog fwd_weight 200"; bash tools/gpu_quick.sh r2h/c3_w200 --config c3_dog --opt ols_fwd_weight=200 --steps 20 --warmup 3
echo "== c3_dog"; bash tools/gpu_quick.sh r2h/c3_dog --config c3_dog --steps 20 --warmup 3
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_ols.log
