#!/bin/bash
# round 6, session b: parity with the tabled rotation of the overlap-save block transforms, A/B against the running product
# (tools/lab/libcwt_rot0.so = -DCWT_OLS_ROT_TABLE=0) and of the serial schedule, microbenchmark: what does the data cost?
export TMPDIR=/tmp
OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print("%s ms %.4f idle %.4f kernels %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0), {n:round(v["ms_per_step"]*1e3,1) for n,v in k.items()}))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  for lib in rot1 rot0; do for s in 0 1 2; do
    L=""; [ $lib = rot0 ] && L="--lib tools/lab/libcwt_rot0.so"
    f=$OUT/c2_${lib}_s${s}_$rep.json
    timeout 300 $B $L --opt serial_rows=$s --detail $f > /dev/null 2> $OUT/err.txt; line $f
  done; done
done
for c in c3_paul c3_dog paul64; do for rep in 1 2; do for lib in rot1 rot0; do for s in 0 1; do
  L=""; [ $lib = rot0 ] && L="--lib tools/lab/libcwt_rot0.so"
  f=$OUT/${c}_${lib}_s${s}_$rep.json
  timeout 300 $B $L --config $c --opt serial_rows=$s --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done; done
timeout 600 tools/lab/stream_poly4 data > $OUT/stream_poly4_data.txt 2>&1; echo "microbench rc=$?"
timeout 300 tools/lab/stream_poly3 > $OUT/stream_poly3.txt 2>&1
for s in 1 2; do
  P=$PWD/$OUT/trace_s$s; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt serial_rows=$s > $P/log.txt 2>&1
  python tools/timeline.py $P --steps 2 --steady > $OUT/timeline_s$s.txt 2>&1
  find $P -type f -size +8M -delete
done
echo done
