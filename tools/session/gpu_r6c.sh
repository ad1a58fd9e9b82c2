#!/bin/bash
# round 6, session c: the serial schedule after its fixes (rows of the band-passed signal last among the block transforms, its
# second pass on 256-thread workgroups), serial_rows = 2 (first block spectra on the caller's stream, FFT aside), ordering
# events without timestamps against CWT_EVENT_TIMING=1
export TMPDIR=/tmp
OUT=gpurun_out/r6c; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s ms %.4f idle %.4f" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  for et in 0 1; do for s in 0 1 2; do
    f=$OUT/c2_et${et}_s${s}_$rep.json
    CWT_EVENT_TIMING=$et timeout 300 $B --opt serial_rows=$s --detail $f > /dev/null 2> $OUT/err.txt; line $f
  done; done
  f=$OUT/c2_s1_bigb_$rep.json
  timeout 300 $B --opt serial_rows=1 --opt aols_small_b=0 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done
for c in c3_paul c3_dog paul64; do for rep in 1 2; do for s in 0 1 2; do
  f=$OUT/${c}_s${s}_$rep.json
  timeout 300 $B --config $c --opt serial_rows=$s --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for s in 1 2; do
  P=$PWD/$OUT/trace_s$s; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$s > $P/log.txt 2>&1
  python tools/timeline.py $P --steps 2 --steady > $OUT/timeline_s$s.txt 2>&1
  find $P -type f -size +8M -delete
done
echo done
