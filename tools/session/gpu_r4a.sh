#!/bin/bash
# round 4, session a: first look at the rows clipped at Nyquist as overlap-save rows on the band-passed signal (aols)
export TMPDIR=/tmp
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout 120 tools/microbench/stream_fma > $OUT/stream_fma.txt 2>&1; cat $OUT/stream_fma.txt
timeout 300 python /root/repo/tools/session/r4a_check.py > $OUT/check.txt 2>&1; tail -30 $OUT/check.txt
for v in "" "--opt aols=0"; do
  bash tools/gpu_quick.sh r4a/c2_$(echo $v | tr -d ' =-') $v
done
for c in c3_paul c3_dog; do
  bash tools/gpu_quick.sh r4a/${c} --config $c
  bash tools/gpu_quick.sh r4a/${c}_noaols --config $c --opt aols=0
done
bash tools/gpu_quick.sh r4a/c2_t16 --opt tolerance_neglog10=16
bash tools/gpu_quick.sh r4a/c2_t16_noaols --opt tolerance_neglog10=16 --opt aols=0
