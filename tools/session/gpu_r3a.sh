#!/bin/bash
# round 3, session a: tolerance sweeps, timing-only ablations (filter evaluation, running-product twiddles)
export TMPDIR=/tmp
OUT=gpurun_out/r3a
mkdir -p $OUT
echo "== c2 default"; bash tools/gpu_quick.sh r3a/c2 --steps 20 --warmup 3
echo "== c2 tol 1e-16"; bash tools/gpu_quick.sh r3a/c2_t16 --steps 20 --warmup 3 --opt tolerance_neglog10=16
echo "== c2 ols_big"; bash tools/gpu_quick.sh r3a/c2_big --steps 20 --warmup 3 --opt ols_big=1
echo "== c2 ols_big 1024"; bash tools/gpu_quick.sh r3a/c2_big1024 --steps 20 --warmup 3 --opt ols_big=1 --opt ols_big_min_halo=1024
timeout 600 python tests/perf/tolerance_sweep.py --config c2 > $OUT/tol_c2.txt 2>&1; tail -12 $OUT/tol_c2.txt
timeout 600 python tests/perf/tolerance_sweep.py --config c3_dog --tol 1e-6,5e-6,1e-5,3e-5,1e-4 > $OUT/tol_c3_dog.txt 2>&1; tail -12 $OUT/tol_c3_dog.txt
timeout 600 python tests/perf/tolerance_sweep.py --config c3_paul --tol 1e-6,5e-6,1e-5,3e-5,1e-4 > $OUT/tol_c3_paul.txt 2>&1; tail -12 $OUT/tol_c3_paul.txt
bash tools/gpu_variants.sh r3a/var "--steps 20 --warmup 3" abl_exp abl_tw
bash tools/gpu_variants.sh r3a/var_dog "--steps 20 --warmup 3 --config c3_dog" abl_exp abl_tw
echo "== c3"; bash tools/gpu_quick.sh r3a/c3_dog --steps 20 --warmup 3 --config c3_dog; bash tools/gpu_quick.sh r3a/c3_paul --steps 20 --warmup 3 --config c3_paul
