#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2e
mkdir -p $OUT
for i in 1 2; do
echo "== c2 default"; bash tools/gpu_quick.sh r2e/c2_$i --steps 20 --warmup 3
echo "== c2 ols_big=0"; bash tools/gpu_quick.sh r2e/c2_big0_$i --opt ols_big=0 --steps 20 --warmup 3
done
echo "== c2 big_min_halo=512"; bash tools/gpu_quick.sh r2e/c2_bmh512 --opt ols_big_min_halo=512 --steps 20 --warmup 3
echo "== c2 big_min_halo=1024"; bash tools/gpu_quick.sh r2e/c2_bmh1024 --opt ols_big_min_halo=1024 --steps 20 --warmup 3
echo "== c2 big_min_halo=128"; bash tools/gpu_quick.sh r2e/c2_bmh128 --opt ols_big_min_halo=128 --steps 20 --warmup 3
python tools/ols_sweep.py --prec 64 > $OUT/ols_sweep_fp64.txt 2>&1; cat $OUT/ols_sweep_fp64.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(tag,d):
    r=d["roofline"]; p=d.get("parity",{})
    print(tag,"value %.1f ms %.3f dom=%s whole=%.3f"%(d["value"],d["ms_per_step"],r["kernel"],r["whole_path"]["frac"]),{k:(round(v["ms_per_step"],3),v["launches_per_step"]) for k,v in r["kernels"].items()},r["row_split"],"parity",p.get("ok"),p.get("max_row_err"),p.get("worst_row"))
    for c,v in sorted(p.get("per_kernel_class",{}).items()):
        if c.startswith("ols"): print("   ",c,v)
show("c2",d)
for k,v in d.get("extra",{}).items(): show(k,v)
PY
