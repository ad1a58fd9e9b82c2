#!/bin/bash
# overlap-save block transforms: padded exchange layout for TB <= 4 (product) / 8 (pad3) / 16 (pad4) residues per tile
# (PMC: 16-36 % of the LDS cycles of the half-tile kernels are bank conflicts); interleaved repeats
export TMPDIR=/tmp
cp pycwt_amd/libcwt_hip.so /tmp/keep.so
for i in 1 2 3; do
for v in product pad3 pad4; do
  [ $v = product ] && cp /tmp/keep.so pycwt_amd/libcwt_hip.so || cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  for c in c2 c3_dog c3_paul; do echo "== $v $c"; bash tools/gpu_quick.sh r3aw/${v}_${c}_$i --config $c --steps 200 --warmup 5 | sed -E "s/dom=.*kernels=/k=/; s/split=.*//" | grep "^value" | sed -E "s/'fwd_pass_a'.*'ols_fwd'/'ols_fwd'/" | cut -c1-150; done
done
done
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
