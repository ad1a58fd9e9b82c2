#!/bin/bash
# round 5, session i: why is the pipelined mode slow -- host time to queue a step, one lane of scratch instead of two
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5i; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f host-enqueue %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0), d.get("host_enqueue_ms_per_step",-1)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
q base
q one_np --opt pipeline=1 --opt pipe_map=2 --opt pipe_prio=0
q one_np_1lane --opt pipeline=1 --opt pipe_map=2 --opt pipe_prio=0 --opt pipe_one_lane=1
q sep_np_1lane --opt pipeline=1 --opt pipe_map=0 --opt pipe_prio=0 --opt pipe_one_lane=1
q chain_1lane --opt pipeline=1 --opt pipe_map=1 --opt pipe_one_lane=1
q base_s3 --shard 3/8
q sep_np_s3 --shard 3/8 --opt pipeline=1 --opt pipe_map=0 --opt pipe_prio=0
q sep_np_1lane_s3 --shard 3/8 --opt pipeline=1 --opt pipe_map=0 --opt pipe_prio=0 --opt pipe_one_lane=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/tr.json --opt pipeline=1 --opt pipe_map=2 --opt pipe_prio=0 --opt pipe_one_lane=1 > $OUT/tr.log 2>&1)
find $OUT -type f -size +6M -delete
