#!/bin/bash
# round 6, session r: k_poly_rows with a wavefront's passes side by side (one LDS read of a coefficient serves every output of a lane)
# against the kernel before (tools/lab/libcwt_polyold.so), interleaved on one box; per (K', degree) class; parity
# (the -D variants / diagnostics of this session were not kept: EXPERIMENTS.md R6.10-R6.12)
export TMPDIR=/tmp
OUT=gpurun_out/r6r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or round4 or chunks or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
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
for rep in 1 2 3; do for v in old new; do
  L=""; [ $v = old ] && L="--lib tools/lab/libcwt_polyold.so"
  f=$OUT/c2_${v}_$rep.json
  timeout 300 $B --config c2 $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog c3_paul paul64 dog64; do for v in old new; do
  L=""; [ $v = old ] && L="--lib tools/lab/libcwt_polyold.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for v in old new; do
  L=""; [ $v = old ] && L="tools/lab/libcwt_polyold.so"
  CWT_LIB=$L timeout 300 python tests/perf/poly_classes.py morlet 64 1e-9 > $OUT/poly_classes_$v.txt 2>&1; echo "-- $v"; grep poly $OUT/poly_classes_$v.txt
done
echo done
