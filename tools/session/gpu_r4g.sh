#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4g; mkdir -p $OUT
SER="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ser -o cwt -- $SER > $OUT/trace_ser.log 2>&1
find $OUT/trace_ser -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serialized.csv \;
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r4g/kernel_stats_serialized.csv")):
    n = r["Name"].replace("void cwt::","")[:70]
    print(f"{n:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
find $OUT -type f -size +4M -delete
for d in 10 12 14; do bash tools/gpu_quick.sh r4g/c2_deg$d --opt poly_degree=$d; done
bash tools/gpu_quick.sh r4g/c3_paul_deg6 --config c3_paul --opt poly_degree=6
bash tools/gpu_quick.sh r4g/c3_paul_deg12 --config c3_paul --opt poly_degree=12
