#!/bin/bash
# A/B: k_poly_rows<double> compiled for 8 waves per SIMD (64 VGPRs, 32 spilled) against the default (70 VGPRs, 7 waves)
export TMPDIR=/tmp
OUT=gpurun_out/r4ac; mkdir -p $OUT
cp pycwt_amd/libcwt_hip.so /tmp/lib_default.so
run() { timeout 120 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), 'poly', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
  cp /tmp/lib_default.so pycwt_amd/libcwt_hip.so; run default
  cp tools/lab/libcwt_w8.so pycwt_amd/libcwt_hip.so; run waves8
done | tee $OUT/ab.txt
cp /tmp/lib_default.so pycwt_amd/libcwt_hip.so
