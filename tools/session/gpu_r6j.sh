#!/bin/bash
# round 6, session j: Chebyshev-economised weights of the polynomial rows (option poly_cheb): every-row parity, A/B interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or round4 or chunks or tolerance_on_gpu or automatic" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    k=d["roofline"].get("kernels",{})
    print("%s ms %.4f idle %.4f | %s | coef %.1f us" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), 1e3*k.get("poly_coef",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for k in 0 1; do
  f=$OUT/c2_ch${k}_$rep.json
  timeout 300 $B --config c2 --opt poly_cheb=$k --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in paul64 c3_paul c3_dog; do for k in 0 1; do
  f=$OUT/${c}_ch${k}_$rep.json
  timeout 300 $B --config $c --opt poly_cheb=$k --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-live-traffic --detail $OUT/c2_full.json > $OUT/c2_full_line.json 2> $OUT/err.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6j/c2_full_line.json").read().strip().splitlines()[-1])
print("c2 with parity:", d["ms_per_step"], d["parity"])
PY
echo done
