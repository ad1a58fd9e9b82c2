#!/bin/bash
# the other mothers in fp64 at the bench target (1e-9): Paul keeps ~100 two-pass rows; does ols_big help it?
export TMPDIR=/tmp
OUT=gpurun_out/r4af; mkdir -p $OUT
run() { timeout 200 python bench.py --config $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 $2', round(d['ms_per_step'],4), round(d['value'],1), 'GS/s', {k:v for k,v in r['row_split'].items() if v}, {k: round(v['us_per_row'],2) for k,v in r['per_class'].items()})"; }
run dog64 ""
run paul64 ""
run paul64 "--opt ols_big=1"
run paul64 "--opt ols_big=2"
run paul64 "--opt ols_big=2 --opt ols_big4_max_halo=8192"
