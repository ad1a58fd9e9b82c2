#!/bin/bash
# round 6, session aa: the product (scalar data path for the polynomial rows of degree <= 4 in complex128, every degree in complex64,
# complex64 rows of R = 64 with two sets per pass) against the LDS path for every row (tools/lab/libcwt_s0_0.so); parity first
export TMPDIR=/tmp
OUT=gpurun_out/r6aa; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or round4 or chunks or golden or full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
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
for rep in 1 2 3; do for c in c2 c3_dog c3_paul; do for v in lds new; do
  L=""; [ $v = lds ] && L="--lib tools/lab/libcwt_s0_0.so"
  f=$OUT/${c}_${v}_$rep.json
  timeout 300 $B --config $c $L --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
