#!/bin/bash
# round 6, session ac: clipped rows with halos of 512 ... 2048 samples in the second (8192-point) class of the band-passed rows (option
# aols_long, complex128): fp64 Paul, every row against the oracle; A/B interleaved
export TMPDIR=/tmp
OUT=gpurun_out/r6ac; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or round4 or tolerance_on_gpu or automatic" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
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
timeout 600 python bench.py --config paul64 --steps 20 --warmup 5 --no-extra --no-live-traffic --detail $OUT/paul64_parity.json > /dev/null 2> $OUT/err.txt; line $OUT/paul64_parity.json
for rep in 1 2 3; do for v in 0 1; do
  f=$OUT/paul64_l${v}_$rep.json
  timeout 300 $B --config paul64 --opt aols_long=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c2 dog64; do for v in 0 1; do
  f=$OUT/${c}_l${v}_$rep.json
  timeout 300 $B --config $c --opt aols_long=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
