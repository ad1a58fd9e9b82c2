#!/bin/bash
# round 4, session j: default bench line with the live PMC passes and the new blocks; GPU suite
export TMPDIR=/tmp
OUT=gpurun_out/r4j; mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/time.txt; echo "bench rc=$?"; tail -3 $OUT/time.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4j/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("c2 value %.1f ms %.4f eff_warmup %s whole %.3f parity %s" % (d["value"], d["ms_per_step"], d.get("effective_warmup_steps"), r["whole_path_frac"], d["parity"]["max_row_err"]))
print(" traffic", r["traffic"], r["traffic_source"], r.get("traffic_calibration"))
for k,v in r["per_class"].items(): print("  ", k, v["rows"], "%.1f us" % (v["ms_per_step"]*1e3), "%.2f us/row" % v["us_per_row"], "frac %.2f" % v["frac"], "traffic", v["traffic_ratio"])
print(" icwt", d.get("icwt"))
for c,e in d.get("extra",{}).items():
    print(c, {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("value","ms_per_step","ms_per_call_median","ms_per_call_min","max_row_err","sampled_pairs_max_row_err","skipped")})
PY
tail -5 $OUT/bench_default.err
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
