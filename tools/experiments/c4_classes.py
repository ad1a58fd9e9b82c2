"""Per-class kernel times of one slab of BASELINE config 4 (128 signals x 2^16 x 128 scales): GPU only, lab tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from collections import Counter
from pycwt_amd import _hip
slab, N, rows = 128, 1 << 16, 128
lam = 4 * np.pi / (6 + np.sqrt(2 + 36))
s0 = 2 / lam; dj = np.log2(N / s0) / (rows - 1)
sj = s0 * 2 ** (np.arange(rows) * dj)
dev = torch.device("cuda", 0)
X = torch.randn(slab, N, dtype=torch.float64, device=dev)
xh = torch.empty(slab, N, dtype=torch.complex128, device=dev)
W = torch.empty(slab, rows, N, dtype=torch.complex128, device=dev)
opts = {k: int(v) for k, v in (o.split("=") for o in sys.argv[1:])}
for mode in ("spectra", "signals"):
    plan = _hip.Plan(N, 64, max_rows=slab * rows, options=dict(opts, profile=1))
    def f():
        if mode == "spectra":
            plan.fft_rows(X.data_ptr(), False, slab, N, N, xh.data_ptr())
            plan.transform_rows_batch(xh.data_ptr(), slab, N, 0, 6.0, 1.0, sj, W.data_ptr(), N, N)
        else:
            plan.transform_batch(X.data_ptr(), slab, N, N, 0, 6.0, 1.0, sj, xh.data_ptr(), W.data_ptr(), N, N)
    for _ in range(30): f()
    plan.sync(); plan.timings()
    reps = 10
    for _ in range(reps): f()
    tm = plan.timings()
    lab = Counter(l.split("/")[0] + ("/" + l.split("/")[1] if l.startswith("ols") or l.startswith("two") else "") for l in plan.row_classes()[:rows])
    print(mode, opts, "per slab ms:", {k: round(ms / reps, 3) for k, (ms, c) in tm.items()}, "sum", round(sum(ms for ms, c in tm.values()) / reps, 3))
    print("   rows of one signal:", dict(lab))
    plan.close()
