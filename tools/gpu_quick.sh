#!/bin/bash
# usage: tools/gpu_quick.sh <tag> [bench args...]  -- one bench line + per-kernel split
export TMPDIR=/tmp
tag=$1; shift
OUT=gpurun_out/$tag
mkdir -p $OUT
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --detail $OUT/bench.json "$@" > $OUT/bench_line.json 2> $OUT/bench.err
python - "$OUT/bench.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.1f GS/s ms/step %.3f  dom=%s whole_frac=%.3f kernels=%s split=%s" % (d["value"], d["ms_per_step"], r["kernel"], r["whole_path"]["frac"], {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in r["kernels"].items()}, r["row_split"]))
except Exception as e:
    print("failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
