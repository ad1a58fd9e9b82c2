#!/bin/bash
# round 6, session z: k_poly_rows with the scalar data path for rows of degree <= D64 / D32 (complex128 / complex64) and the LDS path above:
# s4_24 (4 / all), s4_4 (4 / 4), s2_8 (2 / 8) against the product (0 / 0), interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6z; mkdir -p $OUT
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
for rep in 1 2 3; do for v in base s4_24 s2_8; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2 3; do for c in c3_dog c3_paul; do for v in base s4_24 s4_4 s2_8; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for rep in 1 2; do for c in paul64 dog64; do for v in base s4_24; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
