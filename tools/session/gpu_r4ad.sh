#!/bin/bash
# A/B of a build that ran the block spectra of the two overlap-save tile sizes side by side (option ols_fwd_split; no gain, code removed: EXPERIMENTS.md I.4)
export TMPDIR=/tmp
OUT=gpurun_out/r4ad; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stream_placement" 2>&1 | tail -2
for rep in 1 2 3; do
for cfg in c2 c3_dog c3_paul; do
for o in 1 0; do
  timeout 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --opt ols_fwd_split=$o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg ols_fwd_split=$o', round(d['ms_per_step'],4), round(d['from_idle']['ms_per_step'],4))"
done; done; done | tee $OUT/ab.txt
