#!/usr/bin/env python3
"""How long does the GPU take to reach its sustained clocks from idle?  ms per step of the BASELINE config-2 transform in
consecutive chunks of 10 steps, starting from an idle device (GPU only).  python tools/clock_ramp.py [--chunks 60]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pycwt_amd import _hip
ap = argparse.ArgumentParser(); ap.add_argument("--chunks", type=int, default=60); ap.add_argument("--idle", type=float, default=2.0)
args = ap.parse_args()
N, rows = 1 << 20, 256
sj = bench.scale_grid(N, 1.0, bench.flambda_of(0, 6.0), rows)
plan = _hip.Plan(N, 64, max_rows=rows)
x = np.random.default_rng(1).standard_normal(N)
xd, xh, W = _hip.DeviceBuffer(N * 8), _hip.DeviceBuffer(N * 16), _hip.DeviceBuffer(rows * N * 16)
xd.upload(plan, x)
plan.transform(xd.ptr, N, 0, 6.0, 1.0, sj, xh.ptr, W.ptr, N, N); plan.sync()       # row table, buffers
time.sleep(args.idle)                                                                # let the device fall back to idle
t_start = time.perf_counter()
out = []
for c in range(args.chunks):
    t0 = time.perf_counter()
    for _ in range(10):
        plan.transform(xd.ptr, N, 0, 6.0, 1.0, sj, xh.ptr, W.ptr, N, N)
    plan.sync()
    out.append(((t0 - t_start) * 1e3, (time.perf_counter() - t0) * 100))
print("# ms since the first launch after %.1f s of idle : ms per step over the next 10 steps" % args.idle)
print("  ".join("%.0f:%.3f" % o for o in out))
