#!/bin/bash
# Full round-end style run: smoke, gpu tests, default bench (with cpu baseline), fp32 configs, profiles.
export TMPDIR=/tmp
tag=${1:-full}
OUT=gpurun_out/$tag
mkdir -p $OUT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --detail $OUT/bench_c2.json > $OUT/bench_c2_line.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
for c in c3_paul c3_dog; do timeout 300 python bench.py --config $c --no-cpu-baseline --detail $OUT/bench_$c.json > $OUT/bench_${c}_line.json 2> $OUT/bench_$c.err; done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], "value %.1f ms %.3f dom=%s frac=%.3f whole=%.3f traffic=%s cpu=%s" % (d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["whole_path"]["frac"], r["traffic"], d.get("cpu_baseline",{}).get("value")))
    except Exception as e: print(f, "failed", e)
PY
bash tools/gpu_profile.sh $tag/prof_c2 > $OUT/prof_c2.log 2>&1
bash tools/gpu_profile.sh $tag/prof_c3_paul --config c3_paul > $OUT/prof_c3_paul.log 2>&1
bash tools/gpu_profile.sh $tag/prof_c3_dog --config c3_dog > $OUT/prof_c3_dog.log 2>&1
find $OUT -type f -size +4M -delete
echo done
