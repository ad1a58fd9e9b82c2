#!/bin/bash
# round 3, session s: timing-only ablations and per-class sweeps re-measured at sustained clocks
export TMPDIR=/tmp
OUT=gpurun_out/r3s
mkdir -p $OUT
echo "== product"; bash tools/gpu_quick.sh r3s/c2 --steps 100 --warmup 5
bash tools/gpu_variants.sh r3s/var "--steps 100 --warmup 5" abl_tw abl_wrap abl_pred abl_olsfft abl_olsst abl_all
echo "== product dog"; bash tools/gpu_quick.sh r3s/dog --steps 100 --warmup 5 --config c3_dog
bash tools/gpu_variants.sh r3s/vard "--steps 100 --warmup 5 --config c3_dog" abl_tw abl_wrap abl_pred abl_all
echo "== narrow sweep fp64"; timeout 300 python tools/narrow_sweep.py > $OUT/narrow_sweep_fp64.txt 2>&1; cat $OUT/narrow_sweep_fp64.txt
echo "== ols sweep fp64"; timeout 300 python tools/ols_sweep.py > $OUT/ols_sweep_fp64.txt 2>&1; cat $OUT/ols_sweep_fp64.txt
echo "== ols sweep fp64, default tile only"; timeout 300 python tools/ols_sweep.py --opt ols_small_max_halo=0 > $OUT/ols_sweep_fp64_h0.txt 2>&1; cat $OUT/ols_sweep_fp64_h0.txt
