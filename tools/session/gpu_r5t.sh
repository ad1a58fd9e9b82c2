#!/bin/bash
# round 5, session t: block spectra of the overlap-save rows early (signal on its own stream), nothing else moved
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5t; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for rep in 1 2 3; do q seq_$rep; q early_$rep --input-stream; done
q dog_seq --config c3_dog; q dog_early --config c3_dog --input-stream
q paul_seq --config c3_paul; q paul_early --config c3_paul --input-stream
q ro_seq --opt tolerance_neglog10=16; 
q s0_seq --shard 0/8; q s0_early --shard 0/8 --input-stream
q s1_seq --shard 1/8; q s1_early --shard 1/8 --input-stream
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --input-stream --detail $OUT/tr.json > $OUT/tr.log 2>&1)
python tools/timeline.py $OUT/tr --steps 2 --steady > $OUT/timeline.txt 2>&1
find $OUT -type f -size +6M -delete
