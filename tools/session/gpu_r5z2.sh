#!/bin/bash
# round 5, session z2: does a plan lose its overlap when older streams already sit on the hardware queues?
export TMPDIR=/tmp
OUT=gpurun_out/r5z2; mkdir -p $OUT
for k in 0 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py --config c3_paul --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --dummy-streams $k --detail $OUT/c3_paul_d$k.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$OUT/c3_paul_d$k.json')); print('c3_paul, $k idle streams first: %.4f ms, sum of kernels %.4f' % (d['ms_per_step'], sum(v['ms_per_step'] for v in d['roofline']['kernels'].values())))"
done
for k in 0 1 2 3; do
  timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --dummy-streams $k --detail $OUT/c2_d$k.json > /dev/null 2>&1
  python -c "import json; d=json.load(open('$OUT/c2_d$k.json')); print('c2, $k idle streams first: %.4f ms, sum of kernels %.4f' % (d['ms_per_step'], sum(v['ms_per_step'] for v in d['roofline']['kernels'].values())))"
done
echo done
