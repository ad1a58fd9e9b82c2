#!/bin/bash
# per-rank share of the strong-scaling workload on ONE GPU (bench.py --shard R/G --force-dist): ms per step of every rank
export TMPDIR=/tmp
OUT=gpurun_out/r4o; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/all.json 2>/dev/null
for G in 2 4 8; do
  for R in $(seq 0 $((G-1))); do
    timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic > $OUT/s_${G}_$R.json 2> $OUT/s_${G}_$R.err
  done
done
python - <<'PY'
import json,glob
a=json.loads(open("gpurun_out/r4o/all.json").read().strip().splitlines()[-1])
print("all 256 rows: %.4f ms" % a["ms_per_step"])
for G in (2,4,8):
    ms=[]; rows=[]
    for R in range(G):
        d=json.loads(open(f"gpurun_out/r4o/s_{G}_{R}.json").read().strip().splitlines()[-1])
        ms.append(d["ms_per_step"]); rows.append(d["roofline"]["row_split"])
    print(f"G={G}: per-rank ms", " ".join(f"{m:.3f}" for m in ms), "-> slowest %.3f, speed-up %.2fx" % (max(ms), a["ms_per_step"]/max(ms)))
    for R,r in enumerate(rows): print("    rank",R,{k:v for k,v in r.items() if v})
PY
