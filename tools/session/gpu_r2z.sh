#!/bin/bash
export TMPDIR=/tmp
echo "== base sweep"; python tools/ols_sweep.py --prec 64 2>&1 | grep "ols/"
cp pycwt_amd/libcwt_hip.so /tmp/keep.so; cp tools/experiments/_variants/abl3.so pycwt_amd/libcwt_hip.so
echo "== abl3 (no band loads) sweep"; python tools/ols_sweep.py --prec 64 2>&1 | grep "ols/"
cp /tmp/keep.so pycwt_amd/libcwt_hip.so
