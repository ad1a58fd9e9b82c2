#!/bin/bash
# usage: tools/gpu_variants.sh <tag> "<bench args>" name1 name2 ...  -- bench each library variant (GPU box scratch copy only)
tag=$1; shift
args=$1; shift
cp pycwt_amd/libcwt_hip.so /tmp/libcwt_hip.keep.so
for v in "$@"; do
  echo "== variant $v [$args]"
  cp tools/experiments/_variants/$v.so pycwt_amd/libcwt_hip.so
  bash tools/gpu_quick.sh $tag/$v $args
done
cp /tmp/libcwt_hip.keep.so pycwt_amd/libcwt_hip.so
