#!/bin/bash
# round 3, session af: overlap-save blocks of four tiles (ols_big = 2) for the K = 2048 / multi-term rows
export TMPDIR=/tmp
OUT=gpurun_out/r3af
mkdir -p $OUT
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3af/$tag --steps 200 --warmup 5 "$@" | sed -E 's/dom=.*kernels=/k=/; s/split=.*//' | cut -c1-330; }
q c2; q c2_b
for m in 4096 5120 6144 8192; do q c2_big4_$m --opt ols_big=2 --opt ols_big4_max_halo=$m; done
q c2_big4_min1536 --opt ols_big=2 --opt ols_big4_max_halo=5120 --opt ols_big4_min_halo=1536
q dog --config c3_dog; q dog_big4 --config c3_dog --opt ols_big=2; q dog_big4_5120 --config c3_dog --opt ols_big=2 --opt ols_big4_max_halo=5120
q paul --config c3_paul; q paul_big4 --config c3_paul --opt ols_big=2
timeout 600 python -m pytest tests -q -m gpu -x -k "overlap_save_rows_on_gpu" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
