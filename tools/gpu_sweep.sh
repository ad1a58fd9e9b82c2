#!/bin/bash
# usage: tools/gpu_sweep.sh <tag> "<bench args 1>" "<bench args 2>" ...  -- one gpu_quick line per variant
tag=$1; shift
i=0
for v in "$@"; do
  echo "== [$i] $v"
  bash tools/gpu_quick.sh $tag/v$i $v
  i=$((i+1))
done
