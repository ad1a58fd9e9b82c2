#!/bin/bash
# round 6, session n: side streams at the highest stream priority (CWT_SIDE_PRIO bit mask: 1 = side 0 (FFT, bands, 16384-point coefficient
# tiles), 2 = side 1 (block spectra, band-passed signal), 4 = side2 (8192- / 4096-point coefficient tiles)), interleaved on one box
# (the -D variants / diagnostics of this session were not kept: EXPERIMENTS.md R6.10-R6.12)
export TMPDIR=/tmp
OUT=gpurun_out/r6n; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s ms %.4f idle %.4f" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0)))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 0 5 7 2; do
  f=$OUT/c2_prio${v}_$rep.json
  CWT_SIDE_PRIO=$v timeout 300 $B --config c2 --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
for rep in 1 2; do for c in c3_dog paul64; do for v in 0 5 7; do
  f=$OUT/${c}_prio${v}_$rep.json
  CWT_SIDE_PRIO=$v timeout 300 $B --config $c --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done; done
for v in 5 7; do
P=$PWD/$OUT/trace_prio$v; mkdir -p $P
CWT_SIDE_PRIO=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic > $P/log.txt 2>&1
python tools/timeline.py $P --steps 1 --steady > $OUT/timeline_prio$v.txt 2>&1
find $P -type f -size +8M -delete
head -20 $OUT/timeline_prio$v.txt
done
echo done
