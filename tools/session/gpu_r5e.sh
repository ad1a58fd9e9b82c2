#!/bin/bash
# round 5, session e: kernel trace of the pipelined mode with 8 hardware queues
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5e; mkdir -p $OUT
for v in "pipe8:--opt pipeline=1 --opt pipe_prio=0" "pipe8prio:--opt pipeline=1"; do
  tag=${v%%:*}; args=${v#*:}
  (cd /tmp && GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o cwt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic --no-prime --detail $OUT/$tag.json $args > $OUT/$tag.log 2>&1)
  find $OUT/$tag -type f -size +6M -delete
done
