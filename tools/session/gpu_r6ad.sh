#!/bin/bash
# round 6, session ad: a lower preferred degree while K' stays small (options poly_degree_small / poly_small_logk): 0 (= 8 everywhere),
# (the options of this session were not kept: EXPERIMENTS.md R6.12)
# 6 up to K' = 4096, 6 up to 2048, 4 up to 2048, interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6ad; mkdir -p $OUT
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
for rep in 1 2 3; do for v in "0 12" "6 12" "6 11" "4 11"; do
  set -- $v
  f=$OUT/c2_d$1_k$2_$rep.json
  timeout 300 $B --config c2 --opt poly_degree_small=$1 --opt poly_small_logk=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in dog64 paul64 c3_dog; do for v in "0 12" "6 11"; do
  set -- $v
  f=$OUT/${c}_d$1_k$2_$rep.json
  timeout 300 $B --config $c --opt poly_degree_small=$1 --opt poly_small_logk=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
echo done
