#!/bin/bash
# round 3, session f: block spectra from the half-length packed transform (ols_fwd_real) x half-size tiles
export TMPDIR=/tmp
OUT=gpurun_out/r3f
mkdir -p $OUT
for r in 1 0; do for h in 512 0 704; do echo "== c2 fwd_real=$r small_max_halo=$h"; bash tools/gpu_quick.sh r3f/c2_r${r}_h$h --steps 30 --warmup 3 --opt ols_small_max_halo=$h --opt ols_fwd_real=$r; done; done
for c in c3_dog c3_paul; do for h in 512 0 704; do echo "== $c small_max_halo=$h"; bash tools/gpu_quick.sh r3f/${c}_h$h --config $c --steps 30 --warmup 3 --opt ols_small_max_halo=$h; done; done
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save or every_row" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_ols.log
