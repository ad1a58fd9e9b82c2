#!/bin/bash
# kernel time distribution of Monte-Carlo draws at BASELINE config 5's grid
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4ae; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mc -- python tests/perf/wct_bench.py 20 0.25 6 > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/r4ae/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    name = re.sub(r"^void cwt::", "", r["Name"]).split("(")[0]
    print(f"{name[:60]:60s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
find $OUT -type f -size +8M -delete
