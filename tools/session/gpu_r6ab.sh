#!/bin/bash
# round 6, session ab: complex64 -- the scalar data path for every degree (the product of that moment) against the LDS path (tools/lab/libcwt_s4_0.so), five
# interleaved repeats per config
export TMPDIR=/tmp
OUT=gpurun_out/r6ab; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    print("%s ms %.4f idle %.4f | %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items())))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3 4 5; do for c in c3_dog c3_paul; do for v in lds scalar; do
  L=""; [ $v = lds ] && L="--lib tools/lab/libcwt_s4_0.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
