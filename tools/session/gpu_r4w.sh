#!/bin/bash
# per-rank shares after the refit of the shard cost model
export TMPDIR=/tmp
OUT=gpurun_out/r4w; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/shard_all.json 2>/dev/null
for G in 2 4 8; do for R in $(seq 0 $((G-1))); do
  timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic > $OUT/shard_${G}_$R.json 2>/dev/null
done; done
python - <<'PY'
import json, glob
d = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
print("all", d("gpurun_out/r4w/shard_all.json")["ms_per_step"])
for G in (2, 4, 8):
    for R in range(G):
        x = d(f"gpurun_out/r4w/shard_{G}_{R}.json")
        print(G, R, round(x["ms_per_step"], 4), {k: v for k, v in x["roofline"]["row_split"].items() if v})
PY
