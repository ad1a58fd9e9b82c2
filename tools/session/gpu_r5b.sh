#!/bin/bash
# round 5, session b: launch-bounds variants of the overlap-save kernels, two signals in flight, hardware queues
export TMPDIR=/tmp
OUT=gpurun_out/r5b; mkdir -p $OUT
q() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/$tag.json "$@" > $OUT/$tag.line 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "ms %.4f idle %.4f" % (d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)), {k:round(v["ms_per_step"]*1e3,1) for k,v in r["kernels"].items()})
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
q base
q lb3h --lib tools/lab/libcwt_lb3h.so
q lb3hb --lib tools/lab/libcwt_lb3hb.so
q base2
q pipe2 --pipeline 2
GPU_MAX_HW_QUEUES=8 q hwq8
GPU_MAX_HW_QUEUES=8 q hwq8_pipe2 --pipeline 2
GPU_MAX_HW_QUEUES=2 q hwq2
