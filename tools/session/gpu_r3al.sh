#!/bin/bash
# same-box A/B: fp32 K <= 512 band-limited rows (8192-point tiles) at 4 (product) / 5 / 6 waves per SIMD
export TMPDIR=/tmp
cp pycwt_amd/libcwt_hip.so /tmp/keep.so
for i in 1 2 3; do
for v in product n13lb6 n13lb5; do
  [ $v = product ] && cp /tmp/keep.so pycwt_amd/libcwt_hip.so || cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  for c in c3_dog c3_paul; do echo "== $v $c"; bash tools/gpu_quick.sh r3al/${v}_${c}_$i --config $c --steps 200 --warmup 5 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-120; done
done
done
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
