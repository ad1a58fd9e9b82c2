#!/bin/bash
# round 6, session p: the caller's stream waits once for side stream 1 (option serial_s1_once) against once per consumer
export TMPDIR=/tmp
OUT=gpurun_out/r6p; mkdir -p $OUT
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
for rep in 1 2 3 4; do for v in 0 1; do
  f=$OUT/c2_o${v}_$rep.json
  timeout 300 $B --config c2 --opt serial_s1_once=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in dog64 paul64; do for v in 0 1; do
  f=$OUT/${c}_o${v}_$rep.json
  timeout 300 $B --config $c --opt serial_s1_once=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
P=$PWD/$OUT/trace; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic > $P/log.txt 2>&1
python tools/timeline.py $P --steps 1 --steady > $OUT/timeline.txt 2>&1
find $P -type f -size +8M -delete
head -22 $OUT/timeline.txt
echo done
