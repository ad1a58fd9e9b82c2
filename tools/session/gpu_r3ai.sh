#!/bin/bash
# fp32 band-limited kernel at 4 (product) / 6 / 8 waves per SIMD, per K and per term count
export TMPDIR=/tmp
cp pycwt_amd/libcwt_hip.so /tmp/keep.so
for v in product nlb6 nlb8; do
  [ $v = product ] || cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  echo "== $v narrow sweep fp32"; python tools/narrow_sweep.py --prec 32 --bands 12,100,200,400,800,1000,1500,2500,3500 2>&1 | grep -v "^#" | cut -c1-90
  for c in c3_dog c3_paul; do echo "== $v $c"; bash tools/gpu_quick.sh r3ai/${v}_$c --config $c --steps 200 --warmup 5 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | cut -c1-330; done
  cp /tmp/keep.so pycwt_amd/libcwt_hip.so
done
