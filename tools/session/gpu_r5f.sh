#!/bin/bash
# round 5, session f: pipelined mode, one in-order chain per row form (pipe_map=1)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5f; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
q base
q chain --opt pipeline=1
q chain2 --opt pipeline=1
q base2
q dog --config c3_dog
q dog_chain --config c3_dog --opt pipeline=1
q paul --config c3_paul
q paul_chain --config c3_paul --opt pipeline=1
q ro --opt tolerance_neglog10=16
q ro_chain --opt tolerance_neglog10=16 --opt pipeline=1
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_chain -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/tr_chain.json --opt pipeline=1 > $OUT/tr_chain.log 2>&1)
find $OUT -type f -size +6M -delete
