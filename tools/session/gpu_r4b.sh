#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 120 tools/microbench/stream_poly > $OUT/stream_poly.txt 2>&1; cat $OUT/stream_poly.txt
