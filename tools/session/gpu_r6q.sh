#!/bin/bash
# round 6, session q: k_poly_rows touching the coefficient sets of the workgroup A places ahead (CWT_POLY_PREFETCH = 896 / 1792 / 3584:
# tools/lab/libcwt_pf*.so) against the product, interleaved on one box; per (K', degree) class for 1792
# (the -D variants / diagnostics of this session were not kept: EXPERIMENTS.md R6.10-R6.12)
export TMPDIR=/tmp
OUT=gpurun_out/r6q; mkdir -p $OUT
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
for rep in 1 2 3; do for v in base pf896 pf1792 pf3584; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for v in base pf1792; do
  L=""; [ $v != base ] && L="tools/lab/libcwt_$v.so"
  CWT_LIB=$L timeout 300 python tests/perf/poly_classes.py morlet 64 1e-9 > $OUT/poly_classes_$v.txt 2>&1; echo "-- $v"; grep poly $OUT/poly_classes_$v.txt
done
for rep in 1 2; do for c in c3_dog paul64; do for v in base pf1792; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
