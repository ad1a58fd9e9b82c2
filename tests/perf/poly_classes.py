"""Time of the polynomial rows per (K', degree) class: the rows of one class of a 256-scale grid transformed alone, with
the plan's HIP-event timers (option profile).   python tests/perf/poly_classes.py [morlet|paul|dog] [precision] [tolerance] [key=value ...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from pycwt_amd import _hip

name = sys.argv[1] if len(sys.argv) > 1 else "paul"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
kind, param = {"morlet": (0, 6.0), "paul": (1, 4.0), "dog": (2, 2.0)}[name]
N = 1 << 20
sj = bench.scale_grid(N, 1.0, bench.flambda_of(kind, param), 256)
lib = _hip.Library(os.path.abspath(os.environ['CWT_LIB'])) if os.environ.get('CWT_LIB') else _hip.load()      # (a -D variant under tools/lab/)
opts = {"tolerance": tol, "profile": 1}
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    opts[k] = int(v)
plan = _hip.Plan(N, prec, max_rows=256, lib=lib, options=opts)
labels = plan.classify(kind, param, 1.0, sj, N)
groups = collections.OrderedDict()
for j, l in enumerate(labels):
    groups.setdefault(l, []).append(j)
dev = torch.device("cuda:0")
real, cplx = (torch.float64, torch.complex128) if prec == 64 else (torch.float32, torch.complex64)
x = torch.randn(N, dtype=real, device=dev)
xh = torch.empty(N, dtype=cplx, device=dev)
W = torch.empty(256, N, dtype=cplx, device=dev)
es = 16 if prec == 64 else 8
for lab, idx in groups.items():
    s = sj[idx]
    f = lambda: plan.transform(x.data_ptr(), N, kind, param, 1.0, s, xh.data_ptr(), W.data_ptr(), N, N)
    f(); f(); plan.sync(); plan.timings()
    for _ in range(5):
        f()
    plan.sync()
    t = plan.timings()
    got = plan.row_classes()
    rows_ms = sum(v[0] for k, v in t.items() if k in ("poly", "ols", "ols_small", "aols", "pass_a", "pass_b", "narrow")) / 5
    prep_ms = sum(v[0] for k, v in t.items() if k in ("poly_coef", "ols_fwd", "aols_pre")) / 5
    print(f"{lab:18s} {len(idx):3d} rows ({collections.Counter(got).most_common(1)[0][0]}): rows {rows_ms * 1e3 / len(idx):6.2f} us/row = "
          f"{len(idx) * N * es / (rows_ms * 1e-3) / 1e12:5.2f} TB/s; preparation {prep_ms * 1e3 / len(idx):5.2f} us/row")
plan.close()
