#!/bin/bash
# POLY_PASSES variants of the product library (built with -DCWT_POLY_PASSES=n), same box, A/B/A
export TMPDIR=/tmp
cp pycwt_amd/libcwt_hip.so /tmp/base.so
for v in base pp1 pp3 pp4 base; do
  if [ $v = base ]; then cp /tmp/base.so pycwt_amd/libcwt_hip.so; else cp tools/session/_variant_$v.so pycwt_amd/libcwt_hip.so; fi
  echo "== $v"; bash tools/gpu_quick.sh r4n/c2_$v --no-live-traffic | cut -c1-60
  bash tools/gpu_quick.sh r4n/dog_$v --config c3_dog --no-live-traffic | cut -c1-60
done
cp /tmp/base.so pycwt_amd/libcwt_hip.so
