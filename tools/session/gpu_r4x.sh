#!/bin/bash
# full GPU suite + default bench after the short-call path and the refitted shard model
export TMPDIR=/tmp
OUT=gpurun_out/r4x; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?"
cat $OUT/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4x/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_path_frac"], d["roofline"]["traffic"])
for k, v in d["extra"].items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_call_median", "ms_per_call_min", "max_row_err")})
print("icwt", d["icwt"]["ms"], "parity", d["parity"]["max_row_err"])
PY
