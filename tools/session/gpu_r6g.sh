#!/bin/bash
# round 6, session g: persistent tile launches of the overlap-save kernels (option tile_persist: tools/prototypes/tile_persist.patch on top
# of this commit -- measured negative and not kept, EXPERIMENTS R6.7) and the carrier of the polynomial
# rows (option poly_carrier): parity of a GPU subset under CWT_TILE_PERSIST=1, then A/B interleaved on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6g; mkdir -p $OUT
CWT_TILE_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_row or overlap_save or round4 or stream or full_size_rows or persistent or config4_shape" > $OUT/pytest_persist.log 2>&1; echo "pytest(persist) rc=$?"; tail -3 $OUT/pytest_persist.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    print("%s ms %.4f idle %s | %s | err %.2e" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step"),
          " ".join("%s %d x %.2f" % (k, v["rows"], v["us_per_row"]) for k,v in pc.items()), d.get("parity",{}).get("max_row_err",-1)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do
  for v in 0 1; do
    f=$OUT/c2_p${v}_$rep.json
    timeout 300 $B --opt tile_persist=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
  done
done
for c in c3_paul c3_dog paul64; do for rep in 1 2; do for v in "0 1" "1 1" "1 0"; do
  set -- $v
  f=$OUT/${c}_p$1_k$2_$rep.json
  timeout 300 $B --config $c --opt tile_persist=$1 --opt poly_carrier=$2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
P=$PWD/$OUT/trace_p1; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt tile_persist=1 > $P/log.txt 2>&1
python tools/timeline.py $P --steps 2 --steady > $OUT/timeline_p1.txt 2>&1
find $P -type f -size +8M -delete
echo done
