#!/bin/bash
# round 5, session z3: the hardware-queue probe (option queue_probe) against idle streams created before the plan
export TMPDIR=/tmp
OUT=gpurun_out/r5z3; mkdir -p $OUT
export CWT_QUEUE_PROBE_VERBOSE=1
for k in 0 1 2 3 5; do for q in 1 0; do
  timeout 300 python bench.py --config c3_paul --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --dummy-streams $k --opt queue_probe=$q --detail $OUT/c3_paul_d${k}_q$q.json > /dev/null 2> $OUT/c3_paul_d${k}_q$q.err
  python -c "import json; d=json.load(open('$OUT/c3_paul_d${k}_q$q.json')); print('c3_paul, $k idle streams first, queue_probe=$q: %.4f ms' % d['ms_per_step'])"
  grep "\[cwt\]" $OUT/c3_paul_d${k}_q$q.err | sort | uniq -c | head -6
done; done
for k in 0 2; do for q in 1 0; do
  timeout 300 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --dummy-streams $k --opt queue_probe=$q --detail $OUT/c2_d${k}_q$q.json > /dev/null 2> $OUT/c2_d${k}_q$q.err
  python -c "import json; d=json.load(open('$OUT/c2_d${k}_q$q.json')); print('c2, $k idle streams first, queue_probe=$q: %.4f ms' % d['ms_per_step'])"
  grep "\[cwt\]" $OUT/c2_d${k}_q$q.err | sort | uniq -c | head -6
done; done
timeout 600 python -m pytest tests -x -q -m gpu -k "every_row or smoke or abi or c_host" 2>&1 | tail -3
echo done
