#!/bin/bash
# fp32 K = 1024 band-limited rows at 8 waves per SIMD (two 1024-thread workgroups per CU): benches + per-K sweep
export TMPDIR=/tmp
echo "== narrow sweep fp32"; python tools/narrow_sweep.py --prec 32 --bands 12,100,400,800,1000,1500,2500 2>&1 | grep -v "^#" | cut -c1-90
for i in 1 2; do
for c in c3_dog c3_paul; do echo "== $c"; bash tools/gpu_quick.sh r3aj/${c}_$i --config $c --steps 200 --warmup 5 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330; done
done
echo "== c3_paul narrow_terms=4"; bash tools/gpu_quick.sh r3aj/paul_t4 --config c3_paul --steps 200 --warmup 5 --opt narrow_terms=4 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330
echo "== c3_paul narrow_terms=12"; bash tools/gpu_quick.sh r3aj/paul_t12 --config c3_paul --steps 200 --warmup 5 --opt narrow_terms=12 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330
echo "== c3_paul narrow_terms=16"; bash tools/gpu_quick.sh r3aj/paul_t16 --config c3_paul --steps 200 --warmup 5 --opt narrow_terms=16 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_row or c3 or float or 32" 2>&1 | tail -3
