#!/bin/bash
# short calls after the one-launch / page-locked path: tests, C boundary microbenchmark, Python breakdown
export TMPDIR=/tmp
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "short_calls or page_locked or nino3 or small_golden or all_lengths or edge_cases or sample_datasets" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 120 tools/microbench/host_latency 504 97 > $OUT/host_latency_504.txt 2>&1
timeout 120 tools/microbench/host_latency 4000 60 > $OUT/host_latency_4000.txt 2>&1
timeout 120 python tests/perf/latency_breakdown.py > $OUT/breakdown.txt 2>&1
timeout 300 python tests/perf/latency_bench.py > $OUT/latency.txt 2>&1
cat $OUT/host_latency_504.txt $OUT/host_latency_4000.txt $OUT/breakdown.txt $OUT/latency.txt
