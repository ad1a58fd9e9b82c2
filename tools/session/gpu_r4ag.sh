#!/bin/bash
# order of the rows inside k_poly_rows: largest K' (largest coefficient sets, produced first) first or last
export TMPDIR=/tmp
OUT=gpurun_out/r4ag; mkdir -p $OUT
run() { timeout 200 python bench.py --config $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', round(d['ms_per_step'],4), round(d['value'],1), 'GS/s  poly us/row', round(r['per_class']['poly']['us_per_row'],2), 'poly ms', round(r['per_class']['poly']['ms_per_step'],4))"; }
for rep in 1 2; do
for o in 1 0; do
run paul64 "--opt poly_last_first=$o"
run c2 "--opt poly_last_first=$o"
run c2 "--opt poly_last_first=$o --opt tolerance_neglog10=16"
run c3_paul "--opt poly_last_first=$o"
done; done | tee $OUT/ab.txt
