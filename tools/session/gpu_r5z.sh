#!/bin/bash
# round 5, session z: why do the config-3 blocks of the default run come out 6-10 % slower than `--config c3_*` alone?  steps per timed region
export TMPDIR=/tmp
OUT=gpurun_out/r5z; mkdir -p $OUT
for c in c3_dog c3_paul c2; do for s in 20 40 100 20; do
  timeout 300 python bench.py --config $c --steps $s --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/${c}_$s.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$OUT/${c}_$s.json')); print('$c steps $s: %.4f ms (from idle %.4f)  host enqueue %.4f' % (d['ms_per_step'], d['from_idle']['ms_per_step'], d['host_enqueue_ms_per_step']))"
done; done
echo done
