#!/bin/bash
# round 6, session y: k_poly_rows with the coefficient sets through the scalar data path (CWT_POLY_SCALAR: tools/lab/libcwt_polys.so; no LDS,
# no barrier, 32 vector registers) against the product, interleaved on one box; every row against the oracle; per class
export TMPDIR=/tmp
OUT=gpurun_out/r6y; mkdir -p $OUT
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
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-live-traffic --lib tools/lab/libcwt_polys.so --detail $OUT/c2_polys_parity.json > /dev/null 2> $OUT/err.txt; line $OUT/c2_polys_parity.json
for rep in 1 2 3; do for v in base polys; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog c3_paul paul64; do for v in base polys; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for v in base polys; do
  L=""; [ $v != base ] && L="tools/lab/libcwt_$v.so"
  CWT_LIB=$L timeout 300 python tests/perf/poly_classes.py morlet 64 1e-9 > $OUT/poly_classes_$v.txt 2>&1; echo "-- $v"; grep "d8\|d4\|K256/d6" $OUT/poly_classes_$v.txt
done
echo done
