#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
echo "== c2 default"; bash tools/gpu_quick.sh r2u/c2_$i --steps 20 --warmup 3 | cut -c1-330
echo "== c2 narrow_row_xcd=1"; bash tools/gpu_quick.sh r2u/c2_rx_$i --steps 20 --warmup 3 --opt narrow_row_xcd=1 | cut -c1-330
done
echo "== c3_dog default"; bash tools/gpu_quick.sh r2u/dog --config c3_dog --steps 20 --warmup 3 | cut -c1-330
echo "== c3_dog narrow_row_xcd=1"; bash tools/gpu_quick.sh r2u/dog_rx --config c3_dog --steps 20 --warmup 3 --opt narrow_row_xcd=1 | cut -c1-330
