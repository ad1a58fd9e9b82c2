#!/bin/bash
# HIP graph replay of the step (option graph = 1) against plain launches: whole workload and the per-rank shares of 8
export TMPDIR=/tmp
OUT=gpurun_out/r4p; mkdir -p $OUT
for g in 0 1 0 1; do bash tools/gpu_quick.sh r4p/c2_g$g --no-live-traffic --opt graph=$g | cut -c1-50; done
for g in 0 1; do bash tools/gpu_quick.sh r4p/dog_g$g --config c3_dog --no-live-traffic --opt graph=$g | cut -c1-50; done
for g in 0 1; do
  for R in 0 1 3 4 7; do
    timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/8 --force-dist --no-cpu-baseline --no-extra --no-live-traffic --opt graph=$g > $OUT/s_${g}_$R.json 2> $OUT/s_${g}_$R.err
    python -c "
import json; d=json.loads(open('$OUT/s_${g}_$R.json').read().strip().splitlines()[-1]); print('graph=$g rank $R/8: %.4f ms' % d['ms_per_step'])" 2>/dev/null || tail -3 $OUT/s_${g}_$R.err
  done
done
timeout 300 python -m pytest tests -x -q -m gpu -k "stream_overlap or every_row" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
