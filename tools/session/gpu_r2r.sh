#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
echo "== c2 default"; bash tools/gpu_quick.sh r2r/c2_$i --steps 20 --warmup 3 | cut -c1-330
echo "== c2 k2048_tile=8192"; bash tools/gpu_quick.sh r2r/c2_k8_$i --steps 20 --warmup 3 --opt k2048_tile=8192 | cut -c1-330
done
echo "== terms sweep k2048 on 8192 tiles"; python tools/terms_sweep.py 2>&1 | tail -12
cp pycwt_amd/libcwt_hip.so /tmp/keep.so; cp tools/experiments/_variants/k2048nt.so pycwt_amd/libcwt_hip.so
echo "== variant nt stores: c2 k2048_tile=8192"; bash tools/gpu_quick.sh r2r/c2_k8_nt --steps 20 --warmup 3 --opt k2048_tile=8192 | cut -c1-330
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
