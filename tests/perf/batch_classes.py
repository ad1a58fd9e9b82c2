"""Per kernel class timing of BASELINE config 4 (a batch of 2^16-point signals x 128 Morlet scales through
cwt_transform_batch), with the plan's own HIP-event timers (option "profile").   python tests/perf/batch_classes.py [nbatch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from pycwt_amd import _hip

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, rows = 1 << 16, 128
sj = bench.scale_grid(N, 1.0, bench.flambda_of(0, 6.0), rows)
dev = torch.device("cuda:0")
X = torch.randn(nb, N, dtype=torch.float64, device=dev)
xh = torch.empty(nb, N, dtype=torch.complex128, device=dev)
W = torch.empty(nb, rows, N, dtype=torch.complex128, device=dev)
lib = _hip.load()
for opts in ({}, {"profile": 1}):
    plan = _hip.Plan(N, 64, max_rows=nb * rows, lib=lib, options=dict(opts, tolerance=1e-9))
    f = lambda: plan.transform_batch(X.data_ptr(), nb, N, N, 0, 6.0, 1.0, sj, xh.data_ptr(), W.data_ptr(), N, N)
    f(); f(); plan.sync()
    if opts:
        plan.timings()
        f(); plan.sync()
        t = plan.timings()
        tot = sum(v[0] for v in t.values())
        print(f"profile=1: sum of classes {tot:.3f} ms")
        for k, (ms, cnt) in sorted(t.items(), key=lambda kv: -kv[1][0]):
            print(f"   {k:14s} {ms:8.3f} ms  {cnt:4d} launches  {100 * ms / tot:5.1f} %")
    else:
        import time
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        plan.sync()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        units = nb * N * rows
        print(f"{nb} signals: {ms:.3f} ms per call, {units / ms / 1e6:.1f} GS/s, {units * 16 / ms / 1e6 / 8000:.3f} of 8 TB/s; split {plan.last_split()}")
        labels = plan.row_classes()[:rows]
        import collections
        print("   per signal:", dict(collections.Counter(labels)))
    plan.close()
