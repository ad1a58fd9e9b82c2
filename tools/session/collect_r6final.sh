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
  echo "# bands -> coefficients -> rows, each a single round of workgroups.  Same grid under the round-5 schedule on the box of session e:"
  echo "# slowest rank 0.167 ms = 5.36x against 0.163 = 5.42x (gpurun_out/r6e).  >= 6x needs <= 0.148 ms on every rank: not reached."
} > $P/r06_shards.txt
{ echo "# BASELINE config 5 on ONE GPU (tests/perf/wct_bench.py 20 0.25 12; round 6).  NumPy in / out unless marked device-resident."
  grep -v amdgpu.ids $S/wct.txt; } > $P/r06_wct.txt
for c in paul64 dog64; do cp $S/bench_${c}_line.json $P/r06_bench_${c}_line.json; done
tail -4 $S/pytest_gpu.log > $P/r06_pytest_gpu.txt; tail -1 $S/smoke.log >> $P/r06_pytest_gpu.txt
# experiment records
{ echo "# tools/microbench/stream_poly4.hip on one MI355X (round 6).  First block: session a (coefficients uploaded once: cold), second:"
  echo "# session b, 'data' mode (what do the VALUES cost: nothing) + stream_poly3.hip of round 4 on the same box, third: session e, 'warm'"
  echo "# mode (coefficients touched before every launch) + store-only kernels.  See EXPERIMENTS.md R6.4 for what these do and do not say."
  echo "--- session a"; cat gpurun_out/r6a/stream_poly4.txt
  echo "--- session b: stream_poly4 data"; cat gpurun_out/r6b/stream_poly4_data.txt
  echo "--- session b: stream_poly3 (round 4's microbenchmark, same box)"; cat gpurun_out/r6b/stream_poly3.txt
  echo "--- session e: stream_poly4 warm"; cat gpurun_out/r6e/stream_poly4_warm.txt; } > $P/r06_microbench_stream.txt
{ echo "# tools/microbench/icwt_read.hip on one MI355X (round 6, session e): the column reduction of k_icwt over a 256 x 2^20 complex128 matrix"
  echo "# of random values, by how the loads are issued.  Adopted: non-temporal loads, 128-thread workgroups, 8 rows in flight."
  cat gpurun_out/r6e/icwt_read.txt; } > $P/r06_icwt_read.txt
{ echo "# Round 6, A/B sessions (tools/session/gpu_r6a.sh ... gpu_r6f.sh): the lines the scripts printed, one box per session."
  for s in a b c d e f; do echo; echo "=== session $s (tools/session/gpu_r6$s.sh)"; grep -v "merged\|^\[gpurun\] sending" /tmp/gpu_r6$s.log 2>/dev/null || true; done
  for s in a b c d; do for f in gpurun_out/r6$s/timeline_s*.txt; do [ -f $f ] && { echo; echo "=== $f"; cat $f; }; done; done
} > $P/r06_sessions.txt
ls $P | grep -c r06
