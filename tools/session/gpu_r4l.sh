#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4l; mkdir -p $OUT
bash tools/gpu_quick.sh r4l/c2 --no-live-traffic
bash tools/gpu_quick.sh r4l/c3_paul --config c3_paul --no-live-traffic
bash tools/gpu_quick.sh r4l/c3_dog --config c3_dog --no-live-traffic
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o cwt -- $CMD > $OUT/trace.log 2>&1
python tools/timeline.py $OUT/trace --steps 1 > $OUT/timeline.txt 2>&1; tail -22 $OUT/timeline.txt
find $OUT -type f -size +4M -delete
