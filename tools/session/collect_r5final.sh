#!/bin/bash
# profiles/r05_* from the outputs of tools/session/gpu_r5final.sh (gpurun_out/r5final): one session, one box.
# (r05_pipeline_experiment.txt, r05_chained_experiment.txt come from their own sessions; r05_kernel_resources.txt from the compiler.)
set -e
S=gpurun_out/r5final; P=profiles
cp $S/bench_default.json $P/r05_bench_default.json
cp $S/bench_line.json $P/r05_bench_line.json
python tools/per_class_table.py $P/r05_bench_default.json > $P/r05_per_class.txt
for c in c2 c3_paul c3_dog; do
  cp $S/prof_$c/summary.txt $P/r05_rocprofv3_${c}_summary.txt
  cp $S/prof_$c/traffic.json $P/traffic_$c.json
  cp $S/prof_$c/kernel_stats_serialized.csv $P/r05_kernel_stats_${c}_serialized.csv
done
cp $S/prof_c2/kernel_stats.csv $P/r05_kernel_stats_c2.csv
cp $S/timeline_c2.txt $P/r05_timeline_c2.txt
{ cat $S/shards.txt; echo
  echo "# Rounds 2 / 3 / 4 on their boxes: slowest rank of 8 0.21 / 0.185 / 0.157 ms = 4.9x / 5.4x / 5.63x.  Round 5 tried to take the"
  echo "# fixed cost per rank (the depth of the chain forward FFT -> bands -> coefficients -> rows) out by running the preparation of"
  echo "# call c + 1 beside the rows of call c: slower (0.146-0.199 against 0.133-0.136 ms per rank), EXPERIMENTS.md R5.2 / R5.3."
} > $P/r05_shards.txt
{ echo "# BASELINE config 5 on ONE GPU (tests/perf/wct_bench.py 20 0.25 12; round 5).  NumPy in / out unless marked device-resident."
  echo "# Monte-Carlo: a call = a fixed part (tens of GB of scratch allocated at its first draw and freed at its end, the first draw's"
  echo "# look at the spectra, row tables of an accuracy target the plan has not seen yet) + draws; the script times calls of 2 and 12"
  echo "# draws and reports both parts (a first version divided ONE call by its draws: 55 ... 450 ms per draw depending on what the call"
  echo "# before it had left in the allocator and the table cache)."
  grep -v amdgpu.ids $S/wct.txt; } > $P/r05_wct.txt
for c in paul64 dog64; do cp $S/bench_${c}_line.json $P/r05_bench_${c}_line.json; done
tail -4 $S/pytest_gpu.log > $P/r05_pytest_gpu.txt; tail -1 $S/smoke.log >> $P/r05_pytest_gpu.txt
ls $P | grep -c r05
