#!/bin/bash
# round 5, session d: kernel trace of the pipelined mode
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5d; mkdir -p $OUT
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime"
for v in "pipe:--opt pipeline=1 --opt pipe_prio=0" "pipeprio:--opt pipeline=1" "base:"; do
  tag=${v%%:*}; args=${v#*:}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/$tag.json $args > $OUT/$tag.log 2>&1)
  python tools/timeline.py $OUT/$tag --steps 3 > $OUT/timeline_$tag.txt 2>&1
  find $OUT/$tag -type f -size +6M -delete
done
tail -n 60 $OUT/timeline_pipe.txt
