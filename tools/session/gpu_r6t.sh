#!/bin/bash
# round 6, session t: preferred largest degree of the polynomial rows (option poly_degree: 8 = default, 6, 10), interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6t; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    k=d["roofline"].get("kernels",{})
    print("%s ms %.4f idle %.4f | %s | coef %.1f us" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), 1e3*k.get("poly_coef",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 8 6 10; do
  f=$OUT/c2_d${v}_$rep.json
  timeout 300 $B --config c2 --opt poly_degree=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
echo done
