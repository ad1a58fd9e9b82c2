#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r3g
mkdir -p $OUT
for c in 0 14 16; do echo "== c2 chunk_rows=$c"; bash tools/gpu_quick.sh r3g/c2_c$c --steps 30 --warmup 3 --opt chunk_rows=$c; done
echo "== c3_paul chunk 24 (fp32: 46 rows)"; bash tools/gpu_quick.sh r3g/paul_c24 --config c3_paul --steps 30 --warmup 3 --opt chunk_rows=23
echo "== c3_paul chunk 46"; bash tools/gpu_quick.sh r3g/paul_c46 --config c3_paul --steps 30 --warmup 3 --opt chunk_rows=46
echo "== c3_dog chunk 22"; bash tools/gpu_quick.sh r3g/dog_c22 --config c3_dog --steps 30 --warmup 3 --opt chunk_rows=22
