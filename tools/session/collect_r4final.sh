#!/bin/bash
# profiles/r04_* from the outputs of tools/session/gpu_r4final.sh (gpurun_out/r4final): one session, one box.
# (r04_microbench_stream.txt, r04_experiments.txt, r04_kernel_resources.txt, r04_skip_experiment.txt come from their own sessions.)
set -e
S=gpurun_out/r4final; P=profiles
cp $S/bench_default.json $P/r04_bench_default.json
python tools/per_class_table.py $P/r04_bench_default.json > $P/r04_per_class.txt
for c in c2 c3_paul c3_dog; do
  cp $S/prof_$c/summary.txt $P/r04_rocprofv3_${c}_summary.txt
  cp $S/prof_$c/traffic.json $P/traffic_$c.json
  cp $S/prof_$c/kernel_stats_serialized.csv $P/r04_kernel_stats_${c}_serialized.csv
done
cp $S/prof_c2/kernel_stats.csv $P/r04_kernel_stats_c2.csv
cp $S/timeline_c2.txt $P/r04_timeline_c2.txt
{ cat $S/shards.txt; echo
  echo "# hipGraph replay of a rank's step (option graph = 1, tools/session/gpu_r4p.sh, another box): ranks 0 / 1 / 3 / 4 of 8"
  echo "#   plain launches 0.150 / 0.174 / 0.155 / 0.158 ms, graph replay 0.152 / 0.172 / 0.155 / 0.162 ms: no gain (the fixed cost is"
  echo "#   the depth of the dependent kernel chain -- forward FFT, bands, coefficients, rows -- not launch overhead)."
  echo "# Before the least-squares refit of the shard cost model (another box, all rows 0.9402 ms): ranks of 8 at 0.146-0.170 ms, 5.55x."
} > $P/r04_shards.txt
cat $S/tolerance_c2.txt $S/tolerance_dog.txt $S/tolerance_paul.txt | grep -v amdgpu.ids > $P/r04_tolerance_sweep.txt
{ echo "# Latency of short calls (round 4): (1) the C boundary, tools/microbench/host_latency.cpp"
  cat $S/host_latency_504.txt; echo; cat $S/host_latency_4000.txt; echo
  echo "# (2) the Python call, tests/perf/latency_breakdown.py and tests/perf/latency_bench.py"
  grep -v amdgpu.ids $S/breakdown.txt; grep -v amdgpu.ids $S/latency.txt; echo
  echo "# Start of round 4 (same scripts): pycwt_amd.cwt 504 x 97 = 116 us (round 3: 120); cwt_execute_host 83 us = H2D 12 + two kernels 30"
  echo "# + D2H spectrum 11 + D2H W 21 (56 queued, partly overlapped) + ~25 for the memcpy of W out of the staging buffer (cache-cold)."
  echo "# After caching the per-call grids in the shim: 88 us.  A single fused launch (forward FFT recomputed by every workgroup): no"
  echo "# gain on the GPU (30.4 against 29.9 us), +17 us with the signal read over PCIe by 13 workgroups: not kept (EXPERIMENTS.md 6b)."
} > $P/r04_latency.txt
{ echo "# Per kernel class timing of a batch (BASELINE config 4 shape, 256 of the 1024 signals), tests/perf/batch_classes.py:"
  echo "# the plan's own HIP-event timers (option profile).  Round 3 / start of round 4: 8.61 ms per call, 9 two-pass rows per"
  echo "# signal at 1.49 ms (1.6 TB/s); now those rows run on the band-passed signals (aols + aols_pre)."
  grep -v amdgpu.ids $S/batch_classes.txt; } > $P/r04_batch_classes.txt
{ echo "# BASELINE config 5 on ONE GPU (tests/perf/wct_bench.py 20 0.25 30; NumPy in, NumPy out; round 4)"
  grep -v amdgpu.ids $S/wct.txt; echo
  sed -n '/^# Before/,$p' $P/r04_wct.txt; } > /tmp/r04_wct.txt && mv /tmp/r04_wct.txt $P/r04_wct.txt
ls -la $P | grep r04 | wc -l
