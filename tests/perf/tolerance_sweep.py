#!/usr/bin/env python3
"""Accuracy target of the plan (cwt_plan_set_tolerance) against speed and measured error (GPU only).

    python tests/perf/tolerance_sweep.py [--config c2|c3_paul|c3_dog] [--tol 1e-16,1e-12,...] [--opt k=v]

For every target: the row classification, ms per step of the BASELINE workload (N = 2^20, 256 scales, wall clock around
`steps` calls of cwt_transform, inputs resident) and -- every variant's W kept on the device -- the worst per-row error
max|W - W_oracle| / max|W_oracle| and the relative L2 error against the CPU oracle, which is evaluated once, in groups
of 16 rows.  The table behind the default targets (profiles/r03_tolerance_sweep.txt).
"""
import argparse
import os
import sys
import time
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pycwt_amd import _hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--tol", default="1e-16,1e-12,1e-10,1e-9,1e-8")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rows", type=int, default=256)
ap.add_argument("--check-rows", type=int, default=256, help="rows compared with the oracle (random subset if smaller)")
args = ap.parse_args()
opts = {k: int(v) for k, v in (o.split("=") for o in args.opt)}
kind, param, prec, label = bench.CONFIGS[args.config]
N, rows, dt = 1 << 20, args.rows, 1.0
es = 8 if prec == 64 else 4
x = np.random.default_rng(1234).standard_normal(N)
if prec == 32:
    x = x.astype(np.float32)
sj = bench.scale_grid(N, dt, bench.flambda_of(kind, param), rows)
tols = [float(t) for t in args.tol.split(",")]
xd, xh = _hip.DeviceBuffer(N * es), _hip.DeviceBuffer(N * 2 * es)
runs = []
print(f"# {label}, N = 2^20, {rows} scales, options {opts}; ms per step = wall clock over {args.steps} steps")
for t in tols:
    plan = _hip.Plan(N, prec, max_rows=rows, options=dict(opts, tolerance=t))
    W = _hip.DeviceBuffer(rows * N * 2 * es)
    xd.upload(plan, x)
    for _ in range(3):
        plan.transform(xd.ptr, N, kind, param, dt, sj, xh.ptr, W.ptr, N, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.transform(xd.ptr, N, kind, param, dt, sj, xh.ptr, W.ptr, N, N)
    plan.sync()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    classes = plan.row_classes()
    runs.append({"tol": t, "ms": ms, "classes": classes, "W": W, "plan": plan})
    kinds = Counter(c.split("/")[0] for c in classes)
    print(f"tol {t:8.1e}  {ms:7.3f} ms  {N * rows / ms / 1e6:7.1f} GS/s   " +
          "  ".join(f"{k}:{v}" for k, v in sorted(kinds.items())), flush=True)

sys.path.insert(0, ROOT)
from oracle import cwt_oracle as orc  # noqa: E402  (the checker; nothing timed here)
m = orc.Mother(kind, int(param) if kind else param)
dropped = orc.dropped_rows(sj, dt, m)
order = np.random.default_rng(0).permutation(rows)[:args.check_rows]
stats = [{"worst": 0.0, "row": -1, "num": 0.0, "den": 0.0, "per": {}} for _ in runs]
ctype = np.complex128 if prec == 64 else np.complex64
for g in range(0, len(order), 16):
    idx = np.sort(order[g:g + 16])
    with np.errstate(all="ignore"):
        ref = orc.cwt_rows(x, dt, sj[idx], m)
    for r, st in zip(runs, stats):
        for k, j in enumerate(idx):
            if dropped[j]:
                continue
            got = np.empty(N, dtype=ctype)
            r["plan"].lib.check(r["plan"].lib.cwt_memcpy_d2h(r["plan"].h, got.ctypes.data, r["W"].ptr + int(j) * N * 2 * es,
                                                             got.nbytes))
            den = np.abs(ref[k]).max()
            err = float(np.abs(got - ref[k]).max() / (den if den > 0 else 1.0))
            st["num"] += float(np.sum(np.abs(got - ref[k]) ** 2))
            st["den"] += float(np.sum(np.abs(ref[k]) ** 2))
            c = r["classes"][j].split("/")[0]
            if err > st["per"].get(c, (0.0, -1))[0]:
                st["per"][c] = (err, int(j))
            if err > st["worst"]:
                st["worst"], st["row"] = err, int(j)
print("# measured error against the oracle (rows the reference keeps):")
for r, st in zip(runs, stats):
    print(f"tol {r['tol']:8.1e}  worst row error {st['worst']:9.2e} (row {st['row']:3d})  rel L2 {np.sqrt(st['num'] / max(st['den'], 1e-300)):9.2e}   " +
          "  ".join(f"{k} {v[0]:.1e}@{v[1]}" for k, v in sorted(st["per"].items())), flush=True)
