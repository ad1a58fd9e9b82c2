#!/bin/bash
# round 5, session k: block spectra of the two tile sizes side by side (ols_split), default-tile rows first (ols_order)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5k; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for rep in 1 2; do
q old_$rep --opt ols_split=0
q split_$rep
q split_order_$rep --opt ols_order=1
q order_$rep --opt ols_split=0 --opt ols_order=1
done
q dog_old --config c3_dog --opt ols_split=0
q dog_split --config c3_dog
q dog_split_order --config c3_dog --opt ols_order=1
q paul_old --config c3_paul --opt ols_split=0
q paul_split --config c3_paul
q paul_split_order --config c3_paul --opt ols_order=1
for v in "split:" "split_order:--opt ols_order=1"; do
  tag=${v%%:*}; args=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/tr_$tag.json $args > $OUT/tr_$tag.log 2>&1)
  python tools/timeline.py $OUT/tr_$tag --steps 2 --steady > $OUT/timeline_$tag.txt 2>&1
done
find $OUT -type f -size +6M -delete
