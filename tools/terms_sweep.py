#!/usr/bin/env python3
"""Per-row cost of the single-pass multi-term kernels against the two-pass transform, by filter support (GPU only).

    python tools/terms_sweep.py [--prec 64|32] [--mother 0|1|2]

For a list of supports B (bins) it builds 64 identical rows whose support is about B and times them (HIP events,
option "profile") on each path: two-pass, K = 1024 with ceil(B/1024) aliased terms, K = 2048 (fp64) with
ceil(B/2048) terms.  Prints us/row: the table behind the defaults of "narrow_terms" / "big_terms".
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pycwt_amd import _hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", type=int, default=64)
ap.add_argument("--mother", type=int, default=0)
args = ap.parse_args()
kind = args.mother
param = {0: 6.0, 1: 4.0, 2: 2.0}[kind]
N, rows, dt = 1 << 20, 64, 1.0
es = 8 if args.prec == 64 else 4
x = np.random.default_rng(1).standard_normal(N).astype(np.float64 if es == 8 else np.float32)
xd, xh, W = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es), _hip.DeviceBuffer(rows * N * 2 * es)
paths = {"two_pass": {"narrow_terms": 1, "narrow_big": 0},
         "K1024": {"narrow_terms": 16, "narrow_big": 0}}
if args.prec == 64:
    paths["K2048"] = {"narrow_terms": 1, "big_terms": 8}
# support in bins of a unit-scale filter (s * 2 pi / (N dt) = 1): measured from the plan itself below
print(f"# prec {args.prec} mother {kind}; us per row (64 identical rows, N = 2^20)")
print(f"{'B target':>9s} " + " ".join(f"{p:>22s}" for p in paths))
for target in (2048, 3072, 4096, 6144, 8192, 10240, 12288, 16384):
    line = f"{target:9d} "
    for pname, opts in paths.items():
        plan = _hip.Plan(N, args.prec, max_rows=rows, options=dict(opts, profile=1))
        xd.upload(plan, x)
        plan.forward_fft(xd.ptr, N, xh.ptr)
        # smallest scale whose support still fits T = ceil(target / 1024) terms of 1024 bins (support ~ T * 1024):
        # bisection on the scale with the plan's own classification as the oracle
        T = (target + 1023) // 1024
        lo, hi = 1.0, 1e6
        p2 = _hip.Plan(N, args.prec, max_rows=1, options={"narrow_terms": 16, "narrow_big": 0})
        for _ in range(50):
            mid = np.sqrt(lo * hi)
            p2.transform_rows(xh.ptr, kind, param, dt, np.full(1, mid), W.ptr, N, N)
            c = p2.row_classes()[0]
            t = 99 if c.startswith("two_pass") else (int(c.split("/t")[1]) if "/t" in c else 1)
            if t > T:
                lo = mid            # support too wide: larger scale
            else:
                hi = mid
        p2.close()
        lo = hi
        sj = np.full(rows, lo)
        plan.timings()
        for _ in range(3):
            plan.transform_rows(xh.ptr, kind, param, dt, sj, W.ptr, N, N)
        plan.sync()
        plan.timings()
        reps = 5
        for _ in range(reps):
            plan.transform_rows(xh.ptr, kind, param, dt, sj, W.ptr, N, N)
        tm = plan.timings()
        cls = plan.row_classes()[0]
        us = sum(ms for ms, _ in tm.values()) / reps / rows * 1e3
        plan.close()
        line += f"{us:9.2f} {cls:>12s} "
    print(line, flush=True)
