#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2m
mkdir -p $OUT
python tools/ols_sweep.py --prec 64 > $OUT/ols_sweep_fp64.txt 2>&1; cat $OUT/ols_sweep_fp64.txt
python tools/ols_sweep.py --prec 32 --mother 2 > $OUT/ols_sweep_fp32_dog.txt 2>&1; cat $OUT/ols_sweep_fp32_dog.txt
python tools/ols_sweep.py --prec 32 --mother 1 > $OUT/ols_sweep_fp32_paul.txt 2>&1; cat $OUT/ols_sweep_fp32_paul.txt
timeout 300 python tests/perf/latency_bench.py > $OUT/latency.txt 2>&1; cat $OUT/latency.txt
timeout 300 python tests/perf/config4_bench.py > $OUT/config4.txt 2>&1; cat $OUT/config4.txt
timeout 600 python tests/perf/config5_bench.py > $OUT/config5.txt 2>&1; tail -8 $OUT/config5.txt
for n in 16 17 18 19 21 22; do echo "== logn $n"; bash tools/gpu_quick.sh r2m/n$n --logn $n --steps 20 --warmup 3 | cut -c1-60; done
