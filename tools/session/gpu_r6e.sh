#!/bin/bash
# round 6, session e: the streaming microbenchmark with coefficients as the product leaves them (freshly written), the read
# patterns of k_icwt, every rank's share of config 2 at G = 2, 4, 8 under the serial schedule and under the old one
export TMPDIR=/tmp
OUT=gpurun_out/r6e; mkdir -p $OUT
timeout 600 tools/lab/stream_poly4 warm > $OUT/stream_poly4_warm.txt 2>&1; echo "poly4 rc=$?"
timeout 300 tools/lab/icwt_read > $OUT/icwt_read.txt 2>&1; echo "icwt rc=$?"; cat $OUT/icwt_read.txt
for s in 2 0; do
  D=$OUT/shards_s$s; mkdir -p $D
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$s --detail $D/shard_all.json > /dev/null 2>&1
  for G in 2 4 8; do for R in $(seq 0 $((G-1))); do
    timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$s --detail $D/shard_${G}_$R.json > /dev/null 2>&1
  done; done
  python tools/shard_table.py $D > $OUT/shards_s$s.txt 2>&1; tail -14 $OUT/shards_s$s.txt
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
for c in c3_dog c3_paul c2; do for rep in 1 2 3; do for lib in rot1 rot0; do
  L=""; [ $lib = rot0 ] && L="--lib tools/lab/libcwt_rot0.so"
  f=$OUT/${c}_${lib}_$rep.json
  timeout 300 $B $L --config $c --detail $f > /dev/null 2> $OUT/err.txt
  python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print("%s ms %.4f ols_small %.1f ols %.1f" % (sys.argv[1].split('/')[-1], d["ms_per_step"], k.get("ols_small",{}).get("ms_per_step",0)*1e3, k.get("ols",{}).get("ms_per_step",0)*1e3))
except Exception as e: print(sys.argv[1], "failed", e)
PY
done; done; done
echo done
