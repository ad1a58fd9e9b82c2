#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4k; mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o cwt -- $CMD > $OUT/trace.log 2>&1
python tools/timeline.py $OUT/trace --steps 2 > $OUT/timeline.txt 2>&1; tail -40 $OUT/timeline.txt
find $OUT -type f -size +4M -delete
