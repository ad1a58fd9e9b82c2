#!/bin/bash
# config 5 again after the accuracy target became part of the row tables' key (tests/perf/wct_bench.py), twice
export TMPDIR=/tmp
OUT=gpurun_out/r5final; mkdir -p $OUT
timeout 600 python tests/perf/wct_bench.py 20 0.25 12 > $OUT/wct.txt 2>&1; tail -9 $OUT/wct.txt
timeout 600 python tests/perf/wct_bench.py 20 0.25 12 > $OUT/wct2.txt 2>&1; tail -5 $OUT/wct2.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "coherence or callers or config5 or automatic or tolerance" 2>&1 | tail -3
