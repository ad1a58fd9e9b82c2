#!/usr/bin/env python3
"""Per-row cost of the band-limited kernel by transform length K (GPU only): `rows` identical one-term rows per K.
python tools/narrow_sweep.py [--prec 64|32]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycwt_amd import _hip
ap = argparse.ArgumentParser(); ap.add_argument("--prec", type=int, default=64); ap.add_argument("--rows", type=int, default=48)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--bands", default="12,24,48,100,200,400,800,1000", help="support of the rows in bins, one group of rows per entry")
args = ap.parse_args()
opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
N, rows, dt = 1 << 20, args.rows, 1.0
es = 8 if args.prec == 64 else 4
x = np.random.default_rng(1).standard_normal(N).astype(np.float64 if es == 8 else np.float32)
xd, xh, W = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es), _hip.DeviceBuffer(rows * N * 2 * es)
print(f"# prec {args.prec} opts {opts}: us per row, {rows} identical Morlet rows per scale (support ~ 3.04e6 / s bins)")
for B in [int(b) for b in args.bands.split(",")]:
    s = 3.04e6 / B
    plan = _hip.Plan(N, args.prec, max_rows=rows, options=dict(opts, profile=1))
    xd.upload(plan, x)
    sj = np.full(rows, s)
    t_warm = __import__("time").perf_counter()       # bring the device to its sustained clock first (tools/clock_ramp.py)
    while __import__("time").perf_counter() - t_warm < 0.12:
        plan.transform(xd.ptr, N, 0, 6.0, dt, sj, xh.ptr, W.ptr, N, N)
        plan.sync()
    plan.sync(); plan.timings()
    reps = 20
    for _ in range(reps):
        plan.transform(xd.ptr, N, 0, 6.0, dt, sj, xh.ptr, W.ptr, N, N)
    tm = plan.timings()
    cls = plan.row_classes()[0]
    us = {k: ms / reps / rows * 1e3 for k, (ms, c) in tm.items() if not k.startswith("fwd_")}
    bw = N * 2 * es / (sum(us.values()) * 1e-6) / 1e12
    print(f"B~{B:5d} {cls:16s} " + "  ".join(f"{k} {v:6.2f}" for k, v in us.items()) + f"   {bw:5.2f} TB/s", flush=True)
    plan.close()
