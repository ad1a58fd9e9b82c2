#!/bin/bash
# round 6, session u: price of a block spectrum in the halo-class grouping (option ols_fwd_weight, percent of a row's block transform:
# 100 = default) -- the block spectra of the half-size tiles are the head of the step; and the serial schedule for complex64 again
export TMPDIR=/tmp
OUT=gpurun_out/r6u; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    k=d["roofline"].get("kernels",{})
    print("%s ms %.4f idle %.4f | %s | ols_fwd %.1f us" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), 1e3*k.get("ols_fwd",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 100 200 400 50; do
  f=$OUT/c2_w${v}_$rep.json
  timeout 300 $B --config c2 --opt ols_fwd_weight=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog c3_paul; do for v in 0 2; do
  f=$OUT/${c}_s${v}_$rep.json
  timeout 300 $B --config $c --opt serial_rows=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
