#!/bin/bash
# round 6, session x: the new k_poly_rows at 8 waves per SIMD (lb8), with 4 passes (pp4), with both (pp4lb8) against the product,
# interleaved on one box; kernel traces of pp4 and of the product (what does k_poly_rows take IN the step?)
# (the -D variants of this session were not kept unless EXPERIMENTS.md R6.12 says so)
export TMPDIR=/tmp
OUT=gpurun_out/r6x; mkdir -p $OUT
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
for rep in 1 2 3; do for v in base lb8 pp4 pp4lb8; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for v in base pp4; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  P=$PWD/$OUT/trace_$v; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic $L > $P/log.txt 2>&1
  python tools/timeline.py $P --steps 1 --steady > $OUT/timeline_$v.txt 2>&1
  find $P -type f -size +8M -delete
  echo "-- $v"; grep "k_poly_rows\|steps in the trace" $OUT/timeline_$v.txt
done
for rep in 1 2; do for c in c3_dog c3_paul; do for v in base lb8; do
  L=""; [ $v != base ] && L="--lib tools/lab/libcwt_$v.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
