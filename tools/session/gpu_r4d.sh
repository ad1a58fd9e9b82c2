#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout 120 tools/microbench/stream_poly3 > $OUT/stream_poly3.txt 2>&1; cat $OUT/stream_poly3.txt
