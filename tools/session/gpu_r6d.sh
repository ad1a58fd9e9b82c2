#!/bin/bash
# round 6, session d: serial_rows = 2 with the forward FFT on half-size tiles, serial_rows = 3 (one wait on the caller's stream)
export TMPDIR=/tmp
OUT=gpurun_out/r6d; mkdir -p $OUT
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
for c in c2 c3_paul c3_dog paul64; do for rep in 1 2 3; do
  for v in "0 1" "1 1" "2 0" "2 1" "3 1"; do
    set -- $v
    f=$OUT/${c}_s$1_f$2_$rep.json
    timeout 300 $B --config $c --opt serial_rows=$1 --opt fft_aside_small=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
  done
done; done
for s in 2 3; do
  P=$PWD/$OUT/trace_s$s; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$s > $P/log.txt 2>&1
  python tools/timeline.py $P --steps 3 --steady --anchor "k_ols_fwd_r<double, 11>" > $OUT/timeline_s$s.txt 2>&1
  find $P -type f -size +8M -delete
done
echo done
