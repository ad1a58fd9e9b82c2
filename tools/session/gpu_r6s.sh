#!/bin/bash
# round 6, session s: passes per workgroup of the new k_poly_rows (CWT_POLY_PASSES = 1 / 4: tools/lab/libcwt_pp1.so, pp4.so) against 2
# (the -D variants / diagnostics of this session were not kept: EXPERIMENTS.md R6.10-R6.12)
export TMPDIR=/tmp
OUT=gpurun_out/r6s; mkdir -p $OUT
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
for rep in 1 2 3; do for v in pp2 pp4 pp1; do
  L=""; [ $v != pp2 ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog c3_paul; do for v in pp2 pp4; do
  L=""; [ $v != pp2 ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
