#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2b
mkdir -p $OUT
python tools/ols_sweep.py --prec 64 > $OUT/ols_sweep_fp64.txt 2>&1; cat $OUT/ols_sweep_fp64.txt
python tools/ols_sweep.py --prec 32 --mother 2 > $OUT/ols_sweep_fp32_dog.txt 2>&1; cat $OUT/ols_sweep_fp32_dog.txt
python tools/ols_sweep.py --prec 32 --mother 1 > $OUT/ols_sweep_fp32_paul.txt 2>&1; cat $OUT/ols_sweep_fp32_paul.txt
bash tools/gpu_profile.sh r2b/prof_c2 > $OUT/prof_c2.log 2>&1
grep -n "k_ols\|k_narrow_ct_all<double" $OUT/prof_c2/summary.txt
