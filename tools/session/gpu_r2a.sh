#!/bin/bash
# round-2 session: first contact of the overlap-save rows
export TMPDIR=/tmp
OUT=gpurun_out/r2a
mkdir -p $OUT
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
for cfg in c2 c3_paul c3_dog; do
  echo "== $cfg ols=0"; bash tools/gpu_quick.sh r2a/${cfg}_ols0 --config $cfg --opt ols=0
done
echo "== c2 fwd_weight 50"; bash tools/gpu_quick.sh r2a/c2_w50 --opt ols_fwd_weight=50
echo "== c2 fwd_weight 200"; bash tools/gpu_quick.sh r2a/c2_w200 --opt ols_fwd_weight=200
echo "== c2 max_halo 1024"; bash tools/gpu_quick.sh r2a/c2_h1024 --opt ols_max_halo=1024
echo "== c2 overlap_narrow 0"; bash tools/gpu_quick.sh r2a/c2_on0 --opt overlap_narrow=0
