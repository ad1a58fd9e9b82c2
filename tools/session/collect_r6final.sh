#!/bin/bash
# profiles/r06_* from the outputs of tools/session/gpu_r6final.sh (gpurun_out/r6final): one session, one box; the experiment records
# of sessions a-f (gpurun_out/r6a ... r6f) go to r06_sessions.txt / r06_microbench_stream.txt / r06_icwt_read.txt.
set -e
S=gpurun_out/r6final; P=profiles
cp $S/bench_default.json $P/r06_bench_default.json
cp $S/bench_line.json $P/r06_bench_line.json
cp $S/bench_api_line.json $P/r06_bench_api_line.json
python tools/per_class_table.py $P/r06_bench_default.json > $P/r06_per_class.txt
for c in c2 c3_paul c3_dog; do
  cp $S/prof_$c/summary.txt $P/r06_rocprofv3_${c}_summary.txt
  cp $S/prof_$c/traffic.json $P/traffic_$c.json
  cp $S/prof_$c/kernel_stats_serialized.csv $P/r06_kernel_stats_${c}_serialized.csv
done
cp $S/prof_c2/kernel_stats.csv $P/r06_kernel_stats_c2.csv
cp $S/timeline_c2.txt $P/r06_timeline_c2.txt
{ cat $S/shards.txt; echo
  echo "# Rounds 2 / 3 / 4 / 5 on their boxes: slowest rank of 8 0.21 / 0.185 / 0.157 / 0.164 ms = 4.9x / 5.4x / 5.63x / 5.45x.  Round 6: the"
  echo "# multi-GPU CALL got cheap (pycwt_amd.parallel.cwt_sharded: one collective, no host round trip; bench.py --gpus N times it as"
  echo "# api_ms_per_step), the per-rank fixed cost did not move: 44-46 us = the depth of the chain block spectra -> rows, or forward FFT ->"
  echo "# bands -> coefficients -> rows, each a single round of workgroups; the whole step got faster (0.879 -> 0.86 ms), so the projected"
  echo "# speed-up of 8 ranks did not (5.3-5.5x; the cost model of cwt_plan_balanced_shards was not re-fitted to the faster polynomial rows:"
  echo "# ranks 2 and 4 are the slow ones).  >= 6x needs <= 0.144 ms on every rank: not reached, and unmeasured on a fabric."
} > $P/r06_shards.txt
{ echo "# BASELINE config 5 on ONE GPU (tests/perf/wct_bench.py 20 0.25 12; round 6).  NumPy in / out unless marked device-resident."
  grep -v amdgpu.ids $S/wct.txt; } > $P/r06_wct.txt
for c in paul64 dog64; do cp $S/bench_${c}_line.json $P/r06_bench_${c}_line.json; done
tail -4 $S/pytest_gpu.log > $P/r06_pytest_gpu.txt; tail -1 $S/smoke.log >> $P/r06_pytest_gpu.txt
# (the experiment records -- r06_sessions.txt, r06_microbench_stream.txt, r06_icwt_read.txt -- are not regenerated here: sessions a-f were
# collected by the first version of this script from the session logs, sessions g-u were appended by hand from gpurun_out/r6*/)
ls $P | grep -c r06
