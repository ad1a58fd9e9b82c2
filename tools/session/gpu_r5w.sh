#!/bin/bash
# round 5, session w, x: complex64 overlap-save rows as block PAIRS in packed fp32 (CWT_PAIR_F32) -- parity, then A/B on one box
export TMPDIR=/tmp
OUT=gpurun_out/r5x; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do
for v in default nopair; do
  LIB=""; [ $v != default ] && LIB="--lib gpurun_variants/libcwt_$v.so"
  for c in c3_dog c3_paul; do
    timeout 300 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic $LIB --detail $OUT/${v}_${c}_$rep.json > $OUT/${v}_${c}_$rep.line 2> $OUT/${v}_${c}_$rep.err
    python - <<P
import json
d=json.load(open("$OUT/${v}_${c}_$rep.json"))
pc=d["roofline"]["per_class"]
print("$v $c rep $rep: %.4f ms  %.1f GS/s  " % (d["ms_per_step"], d["value"]) + "  ".join("%s %d rows %.2f us/row" % (k, v["rows"], v["us_per_row"]) for k, v in pc.items()))
P
  done
done
done
echo done
