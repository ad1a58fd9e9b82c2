#!/bin/bash
# round 5, session u: early block spectra + default-tile rows first
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5u; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for rep in 1 2 3; do q seq_$rep; q order_$rep --opt ols_order=1; q early_order_$rep --input-stream --opt ols_order=1; done
q dog_seq --config c3_dog; q dog_eo --config c3_dog --input-stream --opt ols_order=1
q paul_seq --config c3_paul; q paul_eo --config c3_paul --input-stream --opt ols_order=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --input-stream --opt ols_order=1 --detail $OUT/tr.json > $OUT/tr.log 2>&1)
python tools/timeline.py $OUT/tr --steps 2 --steady > $OUT/timeline.txt 2>&1
find $OUT -type f -size +6M -delete
