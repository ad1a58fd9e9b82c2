#!/usr/bin/env python3
"""Per-row cost of pass A and pass B by pass-A class (GPU only): 48 identical rows per support, two-pass forced.
python tools/class_sweep.py [--prec 64|32]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycwt_amd import _hip
ap = argparse.ArgumentParser(); ap.add_argument("--prec", type=int, default=64); ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
N, rows, dt = 1 << 20, 48, 1.0
es = 8 if args.prec == 64 else 4
x = np.random.default_rng(1).standard_normal(N).astype(np.float64 if es == 8 else np.float32)
xd, xh, W = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es), _hip.DeviceBuffer(rows * N * 2 * es)
print(f"# prec {args.prec} opts {opts}: us per row, 48 identical Morlet rows, two-pass forced")
for B in (9000, 14000, 30000, 60000, 120000, 250000, 500000, 1000000):
    s = 2.9 * N / B
    plan = _hip.Plan(N, args.prec, max_rows=rows, options=dict(opts, profile=1, narrow_terms=1, narrow_big=0))
    xd.upload(plan, x)
    plan.forward_fft(xd.ptr, N, xh.ptr)
    sj = np.full(rows, s)
    for _ in range(3):
        plan.transform_rows(xh.ptr, 0, 6.0, dt, sj, W.ptr, N, N)
    plan.sync(); plan.timings()
    reps = 5
    for _ in range(reps):
        plan.transform_rows(xh.ptr, 0, 6.0, dt, sj, W.ptr, N, N)
    tm = plan.timings()
    cls = plan.row_classes()[0]
    print(f"B~{B:8d} {cls:16s} " + "  ".join(f"{k} {ms / reps / rows * 1e3:6.2f}" for k, (ms, c) in tm.items()), flush=True)
    plan.close()
