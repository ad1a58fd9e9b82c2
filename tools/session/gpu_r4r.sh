#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4r; mkdir -p $OUT
bash tools/gpu_quick.sh r4r/c2_base --no-live-traffic | cut -c1-60
bash tools/gpu_quick.sh r4r/c2_nohold --no-live-traffic --opt ols_hold=0 | cut -c1-60
CWT_POLY_PRIORITY=high bash tools/gpu_quick.sh r4r/c2_prio --no-live-traffic | cut -c1-60
CWT_POLY_PRIORITY=high bash tools/gpu_quick.sh r4r/c2_prio_nohold --no-live-traffic --opt ols_hold=0 | cut -c1-60
bash tools/gpu_quick.sh r4r/c2_base2 --no-live-traffic | cut -c1-60
CWT_POLY_PRIORITY=high bash tools/gpu_quick.sh r4r/dog_prio_nohold --config c3_dog --no-live-traffic --opt ols_hold=0 | cut -c1-60
bash tools/gpu_quick.sh r4r/dog_base --config c3_dog --no-live-traffic | cut -c1-60
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --opt ols_hold=0"
CWT_POLY_PRIORITY=high timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o cwt -- $CMD > $OUT/trace.log 2>&1
python tools/timeline.py $OUT/trace --steps 1 > $OUT/timeline.txt 2>&1; tail -20 $OUT/timeline.txt
find $OUT -type f -size +4M -delete
