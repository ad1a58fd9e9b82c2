#!/bin/bash
# default poly_chunk_mb = 96 (balanced chunks) against 0
export TMPDIR=/tmp
OUT=gpurun_out/r4ai; mkdir -p $OUT
run() { timeout 200 python bench.py --config $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', round(d['ms_per_step'],4), round(d['value'],1), 'GS/s  poly ms', round(r['per_class']['poly']['ms_per_step'],4), 'coef ms', round(r['shared_kernels_ms_per_step'].get('poly_coef',0),4))"; }
for rep in 1 2; do
run c2 ""; run c2 "--opt poly_chunk_mb=0"
run paul64 ""; run paul64 "--opt poly_chunk_mb=0"
run dog64 ""; run dog64 "--opt poly_chunk_mb=0"
done | tee $OUT/ab.txt
for o in 96 0; do timeout 300 python tests/perf/tolerance_sweep.py --tol 1e-16 --opt poly_chunk_mb=$o --check-rows 8 2>&1 | grep "^tol" | head -2; done
for o in 96 0; do timeout 300 python tests/perf/tolerance_sweep.py --config c3_paul --tol 1e-8 --opt poly_chunk_mb=$o --check-rows 8 2>&1 | grep "^tol" | head -2; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "every_row or round4 or tolerance_on_gpu or stream_placement" 2>&1 | tail -2
