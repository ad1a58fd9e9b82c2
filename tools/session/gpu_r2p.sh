#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2p
mkdir -p $OUT
for i in 1 2; do echo "== c2"; bash tools/gpu_quick.sh r2p/c2_$i --steps 20 --warmup 3 | cut -c1-330; done
echo "== c3_dog"; bash tools/gpu_quick.sh r2p/c3_dog --config c3_dog --steps 20 --warmup 3 | cut -c1-330
echo "== c3_paul"; bash tools/gpu_quick.sh r2p/c3_paul --config c3_paul --steps 20 --warmup 3 | cut -c1-330
python tools/ols_sweep.py --prec 64 2>&1 | grep "ols/"
timeout 900 python -m pytest tests -q -m gpu -x -k "overlap_save or every_row" > $OUT/pytest_ols.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_ols.log
