"""Does k_poly_rows slow down when the coefficient planes of all rows (150 MB for fp64 Paul at 1e-9) are produced before any
row is written?  The polynomial rows of a grid transformed in one call, and in chunks of bounded coefficient volume.
    python tests/perf/poly_chunks.py [morlet|paul|dog] [precision] [tolerance]"""
import os, sys
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
plan = _hip.Plan(N, prec, max_rows=256, lib=_hip.load(), options={"tolerance": tol, "profile": 1})
labels = plan.classify(kind, param, 1.0, sj, N)
es = 16 if prec == 64 else 8
rows = [(j, int(l.split("/")[1][1:]), int(l.split("/")[2][1:])) for j, l in enumerate(labels) if l.startswith("poly")]
rows.sort(key=lambda r: (r[1], r[2]))
dev = torch.device("cuda:0")
real, cplx = (torch.float64, torch.complex128) if prec == 64 else (torch.float32, torch.complex64)
x = torch.randn(N, dtype=real, device=dev)
xh = torch.empty(N, dtype=cplx, device=dev)
W = torch.empty(256, N, dtype=cplx, device=dev)


def run(idx):
    s = sj[idx]
    f = lambda: plan.transform(x.data_ptr(), N, kind, param, 1.0, s, xh.data_ptr(), W.data_ptr(), N, N)
    f(); f(); plan.sync(); plan.timings()
    for _ in range(5):
        f()
    plan.sync()
    t = plan.timings()
    return t.get("poly", (0, 0))[0] / 5 * 1e3, t.get("poly_coef", (0, 0))[0] / 5 * 1e3


total = sum((d + 1) * k * es for _, k, d in rows)
p_all, c_all = run([j for j, _, _ in rows])
print(f"{name} fp{prec} tol {tol:g}: {len(rows)} polynomial rows, {total / 1e6:.0f} MB of coefficients: k_poly_rows {p_all:.0f} us, coefficient stage {c_all:.0f} us")
for cap in (96e6, 64e6, 32e6):
    chunks, cur, vol = [], [], 0
    for j, k, d in rows:
        b = (d + 1) * k * es
        if cur and vol + b > cap:
            chunks.append(cur); cur, vol = [], 0
        cur.append(j); vol += b
    chunks.append(cur)
    res = [run(c) for c in chunks]
    print(f"  in {len(chunks)} chunks of <= {cap / 1e6:.0f} MB: k_poly_rows {sum(r[0] for r in res):.0f} us, coefficient stage {sum(r[1] for r in res):.0f} us "
          f"({', '.join(f'{len(c)} rows {r[0]:.0f}' for c, r in zip(chunks, res))})")
plan.close()
