#!/bin/bash
# round 6, session l: "serial_rows = 4" of that session (beside the first overlap-save launch only what the polynomial rows need,
# k_poly_rows second, the rest of the preparation beside it) against 2 -- measured negative, not kept (EXPERIMENTS R6.10)
export TMPDIR=/tmp
OUT=gpurun_out/r6l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_placement or every_row" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s ms %.4f idle %.4f" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 2 4; do
  f=$OUT/c2_s${v}_$rep.json
  timeout 300 $B --config c2 --opt serial_rows=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in paul64 dog64; do for v in 2 4; do
  f=$OUT/${c}_s${v}_$rep.json
  timeout 300 $B --config $c --opt serial_rows=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
P=$PWD/$OUT/trace_s4; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=4 > $P/log.txt 2>&1
python tools/timeline.py $P --steps 2 --steady > $OUT/timeline_s4.txt 2>&1
find $P -type f -size +8M -delete
head -40 $OUT/timeline_s4.txt
echo done
