#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2j
mkdir -p $OUT
echo "== shard 0/8 default"; bash tools/gpu_quick.sh r2j/s8 --shard 0/8 --steps 40 --warmup 5
echo "== shard 0/8 no side streams"; bash tools/gpu_quick.sh r2j/s8_ser --shard 0/8 --steps 40 --warmup 5 --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0
echo "== shard 0/8 ols_big=0"; bash tools/gpu_quick.sh r2j/s8_big0 --shard 0/8 --steps 40 --warmup 5 --opt ols_big=0
echo "== shard 0/8 prio0"; CWT_SIDE_PRIORITY=0 bash tools/gpu_quick.sh r2j/s8_p0 --shard 0/8 --steps 40 --warmup 5
echo "== shard 0/8 fwd_weight 300"; bash tools/gpu_quick.sh r2j/s8_w300 --shard 0/8 --steps 40 --warmup 5 --opt ols_fwd_weight=300
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o s8 -- python bench.py --shard 0/8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
python - <<'PY'
import csv,re,glob
f=glob.glob('gpurun_out/r2j/trace/**/s8_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),re.sub(r'\(.*','',r['Kernel_Name']).replace('void cwt::','')[:44],r['Queue_Id']) for r in rows)
i0=60; t0=ev[i0][0]
for s,e,n,q in ev[i0:i0+22]: print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q={q} {n}")
PY
