#!/bin/bash
# round 5, session h: pipelined mode with every row kernel on ONE stream (pipe_map=2)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5h; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
q base
q one --opt pipeline=1 --opt pipe_map=2
q one_np --opt pipeline=1 --opt pipe_map=2 --opt pipe_prio=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_one -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/tr_one.json --opt pipeline=1 --opt pipe_map=2 > $OUT/tr_one.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_one_np -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/tr_one_np.json --opt pipeline=1 --opt pipe_map=2 --opt pipe_prio=0 > $OUT/tr_one_np.log 2>&1)
find $OUT -type f -size +6M -delete
