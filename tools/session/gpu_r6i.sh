#!/bin/bash
# round 6, session i: polynomial rows before the two-pass rows + the carrier (option poly_carrier), A/B interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6i; mkdir -p $OUT
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
for rep in 1 2; do for c in paul64 c3_paul c3_dog c2 dog64; do for k in 0 1; do
  f=$OUT/${c}_k${k}_$rep.json
  timeout 300 $B --config $c --opt poly_carrier=$k --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
