#!/bin/bash
# round 6, session k: compile-time A/B on one box -- k_poly_rows asked for 8 waves per SIMD (tools/lab/libcwt_polylb8.so), wave priority
# of the overlap-save tiles at their stores / first loads (libcwt_priost.so, libcwt_priold.so, libcwt_priost3ld.so) against the product
# (the -D variants / diagnostics of this session were not kept: EXPERIMENTS.md R6.10-R6.12)
export TMPDIR=/tmp
OUT=gpurun_out/r6k; mkdir -p $OUT
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
for rep in 1 2 3; do for v in base polylb8 priost priold priost3ld; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog paul64; do for v in base polylb8 priost; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
