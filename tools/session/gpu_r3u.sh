#!/bin/bash
# round 3, session u: band-limited rows with K <= 128 on the wave-local kernel (k_narrow_wave)
export TMPDIR=/tmp
OUT=gpurun_out/r3u
mkdir -p $OUT
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3u/$tag --steps 100 --warmup 5 "$@" | sed -E 's/dom=.*kernels=/k=/; s/split=.*//' | cut -c1-400; }
q c2_w1; q c2_w0 --opt narrow_wave=0; q c2_w1b; q c2_w0b --opt narrow_wave=0
q dog_w1 --config c3_dog; q dog_w0 --config c3_dog --opt narrow_wave=0
q paul_w1 --config c3_paul; q paul_w0 --config c3_paul --opt narrow_wave=0
echo "== narrow sweep wave=1"; timeout 300 python tools/narrow_sweep.py > $OUT/narrow_sweep_w1.txt 2>&1; cat $OUT/narrow_sweep_w1.txt
echo "== narrow sweep fp32 wave=1"; timeout 300 python tools/narrow_sweep.py --prec 32 > $OUT/narrow_sweep32_w1.txt 2>&1; cat $OUT/narrow_sweep32_w1.txt
echo "== narrow sweep fp32 wave=0"; timeout 300 python tools/narrow_sweep.py --prec 32 --opt narrow_wave=0 > $OUT/narrow_sweep32_w0.txt 2>&1; cat $OUT/narrow_sweep32_w0.txt
timeout 600 python -m pytest tests -q -m gpu -x -k "every_row or all_lengths" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
