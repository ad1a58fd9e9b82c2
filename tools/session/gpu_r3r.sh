#!/bin/bash
# round 3, session r: close calls re-measured at sustained clocks (bench.py now primes the device before the warm-up steps)
export TMPDIR=/tmp
OUT=gpurun_out/r3r
mkdir -p $OUT
q() { tag=$1; shift; echo "== $tag"; bash tools/gpu_quick.sh r3r/$tag --steps 100 --warmup 5 "$@" | cut -c1-46; }
q c2_a; q c2_b
q c2_h0 --opt ols_small_max_halo=0; q c2_h704 --opt ols_small_max_halo=704; q c2_h320 --opt ols_small_max_halo=320
q c2_big1024 --opt ols_big=1 --opt ols_big_min_halo=1024
q c2_ser --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0
q c2_w200 --opt ols_fwd_weight=200
q c2_nt3 --opt narrow_terms=3; q c2_bt8 --opt big_terms=8 --opt narrow_terms=1
q c2_tol8 --opt tolerance_neglog10=8
q dog_a --config c3_dog; q dog_h0 --config c3_dog --opt ols_small_max_halo=0; q dog_ser --config c3_dog --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0
q paul_a --config c3_paul; q paul_h0 --config c3_paul --opt ols_small_max_halo=0; q paul_big0 --config c3_paul --opt ols_big=0
q c2_p2 --pipeline 2
