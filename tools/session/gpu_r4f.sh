#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4f; mkdir -p $OUT
bash tools/gpu_quick.sh r4f/c2
bash tools/gpu_quick.sh r4f/c2_nopoly --opt poly=0
bash tools/gpu_quick.sh r4f/c3_paul --config c3_paul
bash tools/gpu_quick.sh r4f/c3_dog --config c3_dog
bash tools/gpu_quick.sh r4f/c2_t16 --opt tolerance_neglog10=16
