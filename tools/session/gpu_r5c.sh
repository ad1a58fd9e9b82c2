#!/bin/bash
# round 5, session c: the engine's pipelined mode (option "pipeline") against the ordinary call sequence
export TMPDIR=/tmp
OUT=gpurun_out/r5c; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
q base
q pipe --opt pipeline=1
q pipe_noprio --opt pipeline=1 --opt pipe_prio=0
GPU_MAX_HW_QUEUES=8 q pipe_hwq8 --opt pipeline=1
GPU_MAX_HW_QUEUES=8 q pipe_hwq8_noprio --opt pipeline=1 --opt pipe_prio=0
GPU_MAX_HW_QUEUES=8 q base_hwq8
q dog --config c3_dog
GPU_MAX_HW_QUEUES=8 q dog_pipe --config c3_dog --opt pipeline=1
GPU_MAX_HW_QUEUES=8 q paul_pipe --config c3_paul --opt pipeline=1
q paul --config c3_paul
