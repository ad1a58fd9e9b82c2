#!/bin/bash
# round 6, session h: the carrier of the polynomial rows (option poly_carrier) per (K', degree) class, and the every-row parity with it
export TMPDIR=/tmp
OUT=gpurun_out/r6h; mkdir -p $OUT
for k in 0 1; do
  timeout 300 python tests/perf/poly_classes.py paul 64 1e-9 poly_carrier=$k > $OUT/poly_classes_paul64_k$k.txt 2>&1; echo "paul64 carrier $k rc=$?"; cat $OUT/poly_classes_paul64_k$k.txt | grep poly
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or round4 or chunks" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
echo done
