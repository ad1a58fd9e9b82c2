#!/bin/bash
# round 6, session f: parity after the read-stream change (k_icwt, k_time_mean with non-temporal loads) and the tabled rotation in
# the complex64 block pairs; A/B of that rotation at config 3; the "graph" option under the serial schedule
export TMPDIR=/tmp
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
show() { python - $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print("%s ms %.4f idle %.4f ols_small %.1f ols %.1f icwt %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0), k.get("ols_small",{}).get("ms_per_step",0)*1e3, k.get("ols",{}).get("ms_per_step",0)*1e3, d.get("icwt",{}).get("ms")))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for c in c3_dog c3_paul c2; do for rep in 1 2 3; do for lib in rot1 rot0; do
  L=""; [ $lib = rot0 ] && L="--lib tools/lab/libcwt_rot0.so"
  f=$OUT/${c}_${lib}_$rep.json
  timeout 300 $B $L --config $c --detail $f > /dev/null 2> $OUT/err.txt; show $f
done; done; done
for rep in 1 2 3; do for g in 0 1; do
  f=$OUT/c2_graph${g}_$rep.json
  timeout 300 $B --opt graph=$g --detail $f > /dev/null 2> $OUT/err.txt; show $f
done; done
echo done
