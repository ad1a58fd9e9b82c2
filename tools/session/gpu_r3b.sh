#!/bin/bash
# round 3, session b: stream placement (option "sched"), signals in flight (--pipeline), kernel timeline of the default step
export TMPDIR=/tmp
OUT=gpurun_out/r3b
mkdir -p $OUT
for i in 1 2; do echo "== c2 default"; bash tools/gpu_quick.sh r3b/c2_$i --steps 30 --warmup 3; done
for s in 1 2 3; do echo "== c2 sched=$s"; bash tools/gpu_quick.sh r3b/c2_s$s --steps 30 --warmup 3 --opt sched=$s; done
for p in 2 3; do echo "== c2 pipeline=$p"; bash tools/gpu_quick.sh r3b/c2_p$p --steps 30 --warmup 4 --pipeline $p; done
echo "== c2 pipeline=2 sched=3"; bash tools/gpu_quick.sh r3b/c2_p2s3 --steps 30 --warmup 4 --pipeline 2 --opt sched=3
echo "== serialized"; bash tools/gpu_quick.sh r3b/c2_ser --steps 30 --warmup 3 --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o cwt -- $B > $OUT/trace.log 2>&1
python tools/timeline.py $OUT/trace > $OUT/timeline_default.txt; cat $OUT/timeline_default.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_s3 -o cwt -- $B --opt sched=3 > $OUT/trace_s3.log 2>&1
python tools/timeline.py $OUT/trace_s3 > $OUT/timeline_s3.txt; cat $OUT/timeline_s3.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_p2 -o cwt -- $B --pipeline 2 > $OUT/trace_p2.log 2>&1
python tools/timeline.py $OUT/trace_p2 --steps 3 > $OUT/timeline_p2.txt; cat $OUT/timeline_p2.txt
find $OUT -type f -size +4M -delete
