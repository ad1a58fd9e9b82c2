#!/bin/bash
# timing-only experiment: needs a TEMPORARY build with CWT_SKIP hooks (see profiles/r04_skip_experiment.txt; the hooks are not in the sources)
export TMPDIR=/tmp
OUT=gpurun_out/r4z; mkdir -p $OUT
for cfg in c2 c3_dog; do
for skip in "" c o a f coaf P O A POA coafO coafP; do
  CWT_SKIP="$skip" timeout 120 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg skip=[$skip]', round(d['ms_per_step'],4))"
done; done | tee $OUT/skip.txt
