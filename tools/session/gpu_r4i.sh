#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4i; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu -k "round4 or every_row" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash tools/gpu_quick.sh r4i/c3_dog --config c3_dog
bash tools/gpu_quick.sh r4i/c3_dog_noaols --config c3_dog --opt aols=0
bash tools/gpu_quick.sh r4i/c2
