#!/bin/bash
# round 6, session af: block supports of (P/4, P/2] bins through the K = P body of k_ols_ct (option ols_half_full of that session) against two
# aliased P/2-point transforms (the product), interleaved; parity of the variant in the bench's own check
# (the option of this session was not kept unless EXPERIMENTS.md says so)
export TMPDIR=/tmp
OUT=gpurun_out/r6af; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    print("%s ms %.4f idle %.4f | %s | parity %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), d.get("parity",{}).get("max_row_err")))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-live-traffic --opt ols_half_full=1 --detail $OUT/c2_parity.json > /dev/null 2> $OUT/err.txt; line $OUT/c2_parity.json
for rep in 1 2 3; do for v in 0 1; do
  f=$OUT/c2_hf${v}_$rep.json
  timeout 300 $B --config c2 --opt ols_half_full=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in dog64 c3_dog; do for v in 0 1; do
  f=$OUT/${c}_hf${v}_$rep.json
  timeout 300 $B --config $c --opt ols_half_full=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
