#!/bin/bash
# validation of HEAD as the driver will run it: smoke, pytest -m gpu, the bench command
export TMPDIR=/tmp
OUT=gpurun_out/r5head; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_default.json > $OUT/bench_line.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?"; wc -c $OUT/bench_line.json; tail -1 $OUT/bench_line.json | cut -c1-400
