#!/bin/bash
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 600 python -m pytest tests -x -q -m gpu -k "replaced_side_streams or stream_placement or every_row" 2>&1 | tail -2
CWT_QUEUE_PROBE_VERBOSE=1 timeout 200 python bench.py --config c3_paul --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-live-traffic --dummy-streams 2 2>&1 | grep "\[cwt\]\|ms_per_step" | cut -c1-200 | tail -5
