#!/bin/bash
# round 6, session v: per-rank shares at G = 8 after the touch-up of the shard cost model (K' term of the polynomial rows, narrow block supports)
export TMPDIR=/tmp
OUT=gpurun_out/r6v; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_all.json > /dev/null 2>&1
for G in 8; do for R in $(seq 0 $((G-1))); do
  timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_${G}_$R.json > /dev/null 2>&1
done; done
python tools/shard_table.py $OUT 2>&1 | tail -12
