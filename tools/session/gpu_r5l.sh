#!/bin/bash
# round 5, session l: the signal on a stream of its own (chained schedule) against the plan's stream
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5l; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained" 2>&1 | tail -5
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for rep in 1 2; do
q seq_$rep --no-input-stream
q chained_$rep
done
q chained_order --opt ols_order=1
q dog_seq --config c3_dog --no-input-stream
q dog_chained --config c3_dog
q paul_seq --config c3_paul --no-input-stream
q paul_chained --config c3_paul
q s3_seq --shard 3/8 --no-input-stream
q s3_chained --shard 3/8
q s0_seq --shard 0/8 --no-input-stream
q s0_chained --shard 0/8
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/tr.json > $OUT/tr.log 2>&1)
python tools/timeline.py $OUT/tr --steps 2 --steady > $OUT/timeline.txt 2>&1
find $OUT -type f -size +6M -delete
