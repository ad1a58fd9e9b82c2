#!/bin/bash
# round 6, evidence run: everything profiles/r06_* (except the experiment records of sessions a-f) is made from, in ONE session on one box
export TMPDIR=/tmp
OUT=gpurun_out/r6final; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
# the driver's command
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_default.json > $OUT/bench_line.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?"; wc -c $OUT/bench_line.json; cat $OUT/bench_time.txt | tail -3
bash tools/gpu_profile.sh r6final/prof_c2 > $OUT/prof_c2.log 2>&1
for c in c3_paul c3_dog; do
  P=$PWD/gpurun_out/r6final/prof_$c; mkdir -p $P
  SER="python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_ser -o cwt -- $SER > $P/trace_ser.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc_fetch -o cwt -- $SER > $P/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc_write -o cwt -- $SER > $P/pmc_write.log 2>&1
  python tools/summarize_prof.py $P --traffic-json $P/traffic.json > $P/summary.txt 2>&1
  find $P/trace_ser -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats_serialized.csv \;
  find $P -type f -size +8M -delete
done
python tools/timeline.py $PWD/gpurun_out/r6final/prof_c2/trace --steps 3 --steady --anchor "k_ols_fwd_r<double, 11>" > $OUT/timeline_c2.txt 2>&1
# per-rank shares
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_all.json > /dev/null 2>&1
for G in 2 4 8; do for R in $(seq 0 $((G-1))); do
  timeout 120 python bench.py --steps 20 --warmup 3 --shard $R/$G --force-dist --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/shard_${G}_$R.json > /dev/null 2>&1
done; done
python tools/shard_table.py $OUT > $OUT/shards.txt 2>&1; tail -12 $OUT/shards.txt
# the API on one rank with the process group up (what `--gpus N` adds to the line)
timeout 300 python bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-extra --no-live-traffic --detail $OUT/bench_api.json > $OUT/bench_api_line.json 2> $OUT/bench_api.err; tail -c 600 $OUT/bench_api_line.json; echo
timeout 600 python tests/perf/wct_bench.py 20 0.25 12 > $OUT/wct.txt 2>&1; tail -9 $OUT/wct.txt
for c in paul64 dog64; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-extra --no-live-traffic --detail $OUT/bench_$c.json > $OUT/bench_${c}_line.json 2>/dev/null
done
find $OUT -type f -size +8M -delete
echo done
