#!/bin/bash
# round 4, session h: the whole GPU suite + smoke + default bench with the new forms
export TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4h/bench_c2.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("c2 value %.1f ms %.4f from_idle %s whole %.3f parity %s" % (d["value"], d["ms_per_step"], d.get("from_idle",{}).get("ms_per_step"), r["whole_path_frac"], d["parity"]["max_row_err"]))
for k,v in r["per_class"].items(): print("  ", k, v["rows"], "%.1f us" % (v["ms_per_step"]*1e3), "%.2f us/row" % v["us_per_row"], "frac %.2f" % v["frac"])
print("   shared", r["shared_kernels_ms_per_step"])
for c,e in d.get("extra",{}).items():
    print(c, "value %.1f ms %.4f parity %s" % (e["value"], e["ms_per_step"], e["parity"]["max_row_err"]))
PY
