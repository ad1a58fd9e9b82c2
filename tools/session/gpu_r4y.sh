#!/bin/bash
# batches: rows clipped at Nyquist as overlap-save rows on the band-passed signals
export TMPDIR=/tmp
OUT=gpurun_out/r4y; mkdir -p $OUT
timeout 300 python tests/perf/batch_classes.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/batch_classes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "config4 or batch" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
