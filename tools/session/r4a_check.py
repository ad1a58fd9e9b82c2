"""every row of configs 2 / 3 against the oracle on the GPU (rows sampled in groups), with classes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pycwt_amd import _hip
from oracle import cwt_oracle as orc
lib = _hip.load()
N = 1 << 20
for kind, param, prec, tol in ((0, 6.0, 64, 0), (0, 6.0, 64, 1e-16), (1, 4, 32, 0)):
    m = orc.Mother(kind, param)
    s0 = 2 / m.flambda(); dj = np.log2(N / s0) / 255
    sj = s0 * 2 ** (np.arange(256) * dj)
    nr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    x = np.random.default_rng(1234).standard_normal(N)
    plan = _hip.Plan(N, prec, max_rows=nr, lib=lib, options={"tolerance": tol} if tol else {})
    rt = torch.float64 if prec == 64 else torch.float32
    ct = torch.complex128 if prec == 64 else torch.complex64
    xd = torch.from_numpy(x).to("cuda", rt); xh = torch.empty(N, dtype=ct, device="cuda"); W = torch.empty((nr, N), dtype=ct, device="cuda")
    plan.set_stream(torch.cuda.current_stream().cuda_stream)
    plan.transform(xd.data_ptr(), N, kind, param, 1.0, sj[:nr], xh.data_ptr(), W.data_ptr(), N, N)
    torch.cuda.synchronize()
    cl = plan.row_classes()
    bad = orc.dropped_rows(sj[:nr], 1.0, m)
    err = np.zeros(nr)
    for j0 in range(0, nr, 16):
        with np.errstate(all="ignore"):
            ref = orc.cwt_rows(x.astype(np.float32) if prec == 32 else x, 1.0, sj[j0:j0 + 16], m)
        got = W[j0:j0 + 16].cpu().numpy()
        err[j0:j0 + 16] = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print(kind, prec, tol, plan.last_split())
    for j in range(nr):
        if (j % 16 == 0) or (not bad[j] and err[j] > (1e-8 if prec == 64 else 1e-5)):
            print("  row", j, cl[j], "%.2e" % err[j])
    print("  worst", np.nanmax(np.where(bad, 0, err)), "at", int(np.nanargmax(np.where(bad, 0, err))))
    byc = {}
    for j in range(nr):
        if not bad[j]:
            k = cl[j].split("/")[0]; byc[k] = max(byc.get(k, 0), err[j])
    print("  worst per class", {k: "%.2e" % v for k, v in byc.items()})
