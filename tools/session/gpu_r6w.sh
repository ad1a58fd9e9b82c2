#!/bin/bash
# round 6, session w: are the coefficient planes still in the Infinity Cache when k_poly_rows reads them in the step?  poly_chunk_mb
# 96 (one chunk of 57 MB, computed ~300 us before it is read) against 32 / 24 / 16 (2 / 3 / 4 chunks: every chunk but the first is
# computed on the caller's stream right before its rows); kernel trace for 32
export TMPDIR=/tmp
OUT=gpurun_out/r6w; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pc=d["roofline"].get("per_class",{})
    k=d["roofline"].get("kernels",{})
    print("%s ms %.4f idle %.4f | %s | coef %.1f us, poly launches %s" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step",0),
          " ".join("%s %d x %.2f" % (kk, v["rows"], v["us_per_row"]) for kk,v in pc.items()), 1e3*k.get("poly_coef",{}).get("ms_per_step",0), k.get("poly",{}).get("launches_per_step")))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2 3; do for v in 96 32 24 16; do
  f=$OUT/c2_chunk${v}_$rep.json
  timeout 300 $B --config c2 --opt poly_chunk_mb=$v --detail $f > /dev/null 2> $OUT/err.txt; line $f
done; done
P=$PWD/$OUT/trace32; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt poly_chunk_mb=32 > $P/log.txt 2>&1
python tools/timeline.py $P --steps 1 --steady > $OUT/timeline32.txt 2>&1
find $P -type f -size +8M -delete
head -24 $OUT/timeline32.txt
echo done
