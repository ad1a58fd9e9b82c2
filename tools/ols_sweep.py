#!/usr/bin/env python3
"""Per-row cost of the overlap-save rows by scale (GPU only): `rows` identical rows per scale through cwt_transform.
python tools/ols_sweep.py [--prec 64|32] [--mother 0|1|2] [--opt k=v]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycwt_amd import _hip
ap = argparse.ArgumentParser(); ap.add_argument("--prec", type=int, default=64); ap.add_argument("--mother", type=int, default=0)
ap.add_argument("--opt", action="append", default=[]); ap.add_argument("--rows", type=int, default=48)
args = ap.parse_args()
opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
param = {0: 6.0, 1: 4.0, 2: 2.0}[args.mother]
N, rows, dt = 1 << 20, args.rows, 1.0
es = 8 if args.prec == 64 else 4
x = np.random.default_rng(1).standard_normal(N).astype(np.float64 if es == 8 else np.float32)
xd, xh, W = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es), _hip.DeviceBuffer(rows * N * 2 * es)
print(f"# prec {args.prec} mother {args.mother} opts {opts}: us per row, {rows} identical rows per scale, cwt_transform")
for s in (5.0, 6.0, 8.0, 12.0, 16.0, 24.0, 32.0, 48.0, 64.0, 90.0, 128.0, 180.0, 230.0, 300.0, 400.0):
    plan = _hip.Plan(N, args.prec, max_rows=rows, options=dict(opts, profile=1))
    xd.upload(plan, x)
    sj = np.full(rows, s)
    t_warm = __import__("time").perf_counter()       # bring the device to its sustained clock first (tools/clock_ramp.py)
    while __import__("time").perf_counter() - t_warm < 0.12:
        plan.transform(xd.ptr, N, args.mother, param, dt, sj, xh.ptr, W.ptr, N, N)
        plan.sync()
    plan.sync(); plan.timings()
    reps = 20
    for _ in range(reps):
        plan.transform(xd.ptr, N, args.mother, param, dt, sj, xh.ptr, W.ptr, N, N)
    tm = plan.timings()
    cls = plan.row_classes()[0]
    print(f"s={s:7.1f} {cls:16s} " + "  ".join(f"{k} {ms / reps / rows * 1e3:6.2f}" for k, (ms, c) in tm.items()
                                               if not k.startswith("fwd_")), flush=True)
    plan.close()
