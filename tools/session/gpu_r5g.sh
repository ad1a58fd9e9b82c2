#!/bin/bash
# round 5, session g: a rank's share of 8 (bench.py --shard R/8), ordinary calls against the pipelined mode
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5g; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)), d["roofline"].get("row_split"))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for r in 0 3 7; do
  q base_$r --shard $r/8
  q chain_$r --shard $r/8 --opt pipeline=1
  q sep_$r --shard $r/8 --opt pipeline=1 --opt pipe_map=0
  q sepnp_$r --shard $r/8 --opt pipeline=1 --opt pipe_map=0 --opt pipe_prio=0
done
