#!/bin/bash
# round 6, session a: parity of the new schedule / coefficient kernel, A/B of the schedules on one box, the streaming microbenchmark
export TMPDIR=/tmp
OUT=gpurun_out/r6a; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 tools/lab/stream_poly4 > $OUT/stream_poly4.txt 2>&1; echo "microbench rc=$?"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s ms %.4f idle %s split %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step"), d["roofline"].get("row_split")))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  for v in "0 0" "0 1" "1 0" "1 1" "2 1"; do
    set -- $v
    f=$OUT/c2_s$1_c$2_$rep.json
    timeout 300 $B --opt serial_rows=$1 --opt coef_small=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
  done
done
for c in c3_paul c3_dog paul64; do for rep in 1 2; do for v in "0 0" "1 1" "2 1"; do
  set -- $v
  f=$OUT/${c}_s$1_c$2_$rep.json
  timeout 300 $B --config $c --opt serial_rows=$1 --opt coef_small=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for v in "1 1" "2 1"; do
  set -- $v
  P=$PWD/$OUT/trace_s$1; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$1 --opt coef_small=$2 > $P/log.txt 2>&1
  python tools/timeline.py $P --steps 2 --steady > $OUT/timeline_s$1.txt 2>&1
  find $P -type f -size +8M -delete
done
echo done
