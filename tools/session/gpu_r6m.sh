#!/bin/bash
# round 6, session m: rows with halos in (512, 1024] on 8192-point blocks of two half-size tiles (option ols_small_big) instead of the
# default tile: parity, A/B interleaved on one box, kernel trace
export TMPDIR=/tmp
OUT=gpurun_out/r6m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or overlap_save or stream_placement or round4" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    k=d["roofline"].get("kernels",{})
    print("%s ms %.4f idle %.4f | %s | ols_fwd %.1f us" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), 1e3*k.get("ols_fwd",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 0 1; do
  f=$OUT/c2_sb${v}_$rep.json
  timeout 300 $B --config c2 --opt ols_small_big=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in dog64 c3_dog c3_paul paul64; do for v in 0 1; do
  f=$OUT/${c}_sb${v}_$rep.json
  timeout 300 $B --config $c --opt ols_small_big=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
P=$PWD/$OUT/trace; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic > $P/log.txt 2>&1
python tools/timeline.py $P --steps 2 --steady > $OUT/timeline.txt 2>&1
find $P -type f -size +8M -delete
head -36 $OUT/timeline.txt
echo done
