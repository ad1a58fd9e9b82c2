#!/bin/bash
# A/B: polynomial rows in chunks of bounded coefficient volume (option poly_chunk_mb; 0 = all rows at once)
export TMPDIR=/tmp
OUT=gpurun_out/r4ah; mkdir -p $OUT
run() { timeout 200 python bench.py --config $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --opt poly_chunk_mb=$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 chunk_mb=$2', round(d['ms_per_step'],4), round(d['value'],1), 'GS/s  poly ms', round(r['per_class']['poly']['ms_per_step'],4), 'coef ms', round(r['shared_kernels_ms_per_step'].get('poly_coef',0),4), 'parity', d.get('parity',{}).get('max_row_err'))"; }
for rep in 1 2; do
for mb in 0 32 48 64; do
run c2 $mb
done
for mb in 0 48; do
run paul64 $mb
run c3_dog $mb
run c3_paul $mb
done; done | tee $OUT/ab.txt
timeout 300 python tests/perf/tolerance_sweep.py --tol 1e-16 --opt poly_chunk_mb=0 --check-rows 8 2>&1 | grep "^tol" | head -1
timeout 300 python tests/perf/tolerance_sweep.py --tol 1e-16 --opt poly_chunk_mb=48 --check-rows 8 2>&1 | grep "^tol" | head -1
timeout 300 python tests/perf/tolerance_sweep.py --tol 1e-16 --opt poly_chunk_mb=32 --check-rows 8 2>&1 | grep "^tol" | head -1
