#!/bin/bash
# kernel timeline of a step with the current launch order
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4aa; mkdir -p $OUT
for cfg in c2 c3_dog; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$cfg -o cwt -- python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/trace_$cfg.log 2>&1
python tools/timeline.py $OUT/trace_$cfg --steps 2 > $OUT/timeline_$cfg.txt 2>&1
done
find $OUT -type f -size +8M -delete
tail -5 $OUT/timeline_c2.txt
