#!/bin/bash
# where a 504-point call spends its time: C boundary microbenchmark + Python breakdown
export TMPDIR=/tmp
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout 120 tools/microbench/host_latency 504 97 > $OUT/host_latency_504.txt 2>&1
timeout 120 tools/microbench/host_latency 4096 134 > $OUT/host_latency_4096.txt 2>&1
timeout 120 python tests/perf/latency_breakdown.py > $OUT/breakdown.txt 2>&1
timeout 300 python tests/perf/latency_bench.py > $OUT/latency.txt 2>&1
cat $OUT/host_latency_504.txt $OUT/host_latency_4096.txt $OUT/breakdown.txt $OUT/latency.txt
