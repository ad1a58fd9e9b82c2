# device-resident timing: batch of signals N=2^16, 128 rows, one launch set vs per-signal loop
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pycwt_amd import _hip
nb, N, rows = 64, 1 << 16, 128
lam = 4*np.pi/(6+np.sqrt(38)); s0 = 2/lam; dj = np.log2(N/s0)/(rows-1)
sj = s0*2**(np.arange(rows)*dj)
dev = torch.device('cuda', 0)
X = torch.randn(nb, N, dtype=torch.float64, device=dev)
xh = torch.empty(nb, N, dtype=torch.complex128, device=dev)
W = torch.empty(nb, rows, N, dtype=torch.complex128, device=dev)
plan = _hip.Plan(N, 64, max_rows=nb*rows)
plan.set_stream(torch.cuda.current_stream().cuda_stream)
def batched():
    plan.fft_rows(X.data_ptr(), False, nb, N, N, xh.data_ptr())
    plan.transform_rows_batch(xh.data_ptr(), nb, N, 0, 6.0, 1.0, sj, W.data_ptr(), N, N)
def looped():
    for b in range(nb):
        plan.forward_fft(X[b].data_ptr(), N, xh[b].data_ptr())
        plan.transform_rows(xh[b].data_ptr(), 0, 6.0, 1.0, sj, W[b].data_ptr(), N, N)
for name, f in (("batched", batched), ("looped", looped), ("batched", batched)):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    el = (time.perf_counter()-t)/3
    print(f"{name}: {el*1e3:.2f} ms for {nb} signals -> {nb*N*rows/el/1e9:.1f} GSamples*scales/s", plan.last_split())
