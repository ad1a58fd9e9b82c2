#!/bin/bash
# One gpurun call: parity tests, microbenchmarks, first bench line, option sweeps, rocprof stats.
export TMPDIR=/tmp
OUT=gpurun_out/r01a
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | tee -a $OUT/summary.txt
echo "== microbench" | tee -a $OUT/summary.txt
timeout 120 tools/microbench/mem_patterns > $OUT/mem_patterns.log 2>&1; echo "rc=$?" >> $OUT/summary.txt
cat $OUT/mem_patterns.log >> $OUT/summary.txt
echo "== bench default" | tee -a $OUT/summary.txt
timeout 300 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/summary.txt
cat $OUT/bench_default.json >> $OUT/summary.txt; tail -5 $OUT/bench_default.err >> $OUT/summary.txt
for o in "narrow=0" "chunk_rows=1" "chunk_rows=2" "chunk_rows=8" "chunk_rows=16" "wg_points=4096" "narrow_max_k=512" "narrow_max_k=256"; do
  echo "== bench $o" >> $OUT/summary.txt
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --opt $o > $OUT/bench_$o.json 2> $OUT/bench_$o.err
  python - "$OUT/bench_$o.json" >> $OUT/summary.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.1f GS/s ms/step %.3f  dom=%s whole_frac=%.3f kernels=%s split=%s" % (d["value"], d["ms_per_step"], r["kernel"], r["whole_path"]["frac"], {k:round(v["ms_per_step"],3) for k,v in r["kernels"].items()}, r["row_split"]))
except Exception as e:
    print("failed", e)
PY
done
echo "== bench fp32" >> $OUT/summary.txt
for c in c3_paul c3_dog; do
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  cat $OUT/bench_$c.json >> $OUT/summary.txt
done
echo "== rocprof stats" >> $OUT/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cwt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/summary.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" >> $OUT/summary.txt
echo "== pytest gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
# keep the merge small: drop bulky rocprof traces, keep csv summaries
find $OUT/prof -type f ! -name "*.csv" -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
echo done
