#!/bin/bash
# round 4, evidence run: everything profiles/r04_* is made from, in ONE session on one box
export TMPDIR=/tmp
OUT=gpurun_out/r4final; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?"
bash tools/gpu_profile.sh r4final/prof_c2 > $OUT/prof_c2.log 2>&1
for c in c3_paul c3_dog; do
  P=$PWD/gpurun_out/r4final/prof_$c; mkdir -p $P
  SER="python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_ser -o cwt -- $SER > $P/trace_ser.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc_fetch -o cwt -- $SER > $P/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc_write -o cwt -- $SER > $P/pmc_write.log 2>&1
  python tools/summarize_prof.py $P --traffic-json $P/traffic.json > $P/summary.txt 2>&1
  find $P/trace_ser -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats_serialized.csv \;
  find $P -type f -size +8M -delete
done
python tools/timeline.py $PWD/gpurun_out/r4final/prof_c2/trace --steps 2 --steady > $OUT/timeline_c2.txt 2>&1
# per-rank shares
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/shard_all.json 2>/dev/null
for G in 2 4 8; do for R in $(seq 0 $((G-1))); do
  timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic > $OUT/shard_${G}_$R.json 2>/dev/null
done; done
python tools/shard_table.py $OUT > $OUT/shards.txt 2>&1; cat $OUT/shards.txt
timeout 120 tools/microbench/host_latency 504 97 > $OUT/host_latency_504.txt 2>&1
timeout 120 tools/microbench/host_latency 4000 60 > $OUT/host_latency_4000.txt 2>&1
timeout 120 python tests/perf/latency_breakdown.py > $OUT/breakdown.txt 2>&1
timeout 300 python tests/perf/latency_bench.py > $OUT/latency.txt 2>&1; tail -5 $OUT/latency.txt
timeout 300 python tests/perf/batch_classes.py 256 > $OUT/batch_classes.txt 2>&1
timeout 600 python tests/perf/wct_bench.py 20 0.25 30 > $OUT/wct.txt 2>&1
timeout 600 python tests/perf/tolerance_sweep.py --tol 1e-16,1e-12,1e-10,1e-9,1e-8 > $OUT/tolerance_c2.txt 2>&1
timeout 600 python tests/perf/tolerance_sweep.py --config c3_dog --tol 1e-8,1e-6,3e-5,1e-4 > $OUT/tolerance_dog.txt 2>&1
timeout 600 python tests/perf/tolerance_sweep.py --config c3_paul --tol 1e-8,1e-6,3e-5,1e-4 > $OUT/tolerance_paul.txt 2>&1
find $OUT -type f -size +8M -delete
echo done
