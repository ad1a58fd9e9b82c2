#!/bin/bash
# the per-rank shares again (one run of gpu_r5final.sh had a transient on one rank: primed 0.242 ms, from idle 0.165)
export TMPDIR=/tmp
OUT=gpurun_out/r5final; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_all.json > /dev/null 2>&1
for G in 2 4 8; do for R in $(seq 0 $((G-1))); do
  timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_${G}_$R.json > /dev/null 2>&1
done; done
python tools/shard_table.py $OUT > $OUT/shards.txt 2>&1; tail -12 $OUT/shards.txt
