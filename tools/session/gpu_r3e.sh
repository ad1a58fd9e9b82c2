#!/bin/bash
# round 3, session e: overlap-save rows on two tile sizes in one transform; threshold sweep
export TMPDIR=/tmp
OUT=gpurun_out/r3e
mkdir -p $OUT
for h in 704 0 512 640 896; do echo "== c2 small_max_halo=$h"; bash tools/gpu_quick.sh r3e/c2_h$h --steps 30 --warmup 3 --opt ols_small_max_halo=$h; done
for h in 704 0 512 896; do echo "== c3_dog small_max_halo=$h"; bash tools/gpu_quick.sh r3e/dog_h$h --config c3_dog --steps 30 --warmup 3 --opt ols_small_max_halo=$h; done
for h in 704 0 512 896; do echo "== c3_paul small_max_halo=$h"; bash tools/gpu_quick.sh r3e/paul_h$h --config c3_paul --steps 30 --warmup 3 --opt ols_small_max_halo=$h; done
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save or every_row" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_ols.log
