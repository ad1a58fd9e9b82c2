#!/bin/bash
# same-box A/B: fp32 K = 1024 band-limited rows at 8 (product) against 4 (old4) waves per SIMD
export TMPDIR=/tmp
cp pycwt_amd/libcwt_hip.so /tmp/keep.so
for i in 1 2 3; do
for v in product old4; do
  [ $v = product ] && cp /tmp/keep.so pycwt_amd/libcwt_hip.so || cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  for c in c3_dog c3_paul; do echo "== $v $c"; bash tools/gpu_quick.sh r3ak/${v}_${c}_$i --config $c --steps 200 --warmup 5 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-120; done
done
done
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
