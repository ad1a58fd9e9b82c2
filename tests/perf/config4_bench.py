"""BASELINE config 4 on ONE GPU: 1024 signals x N = 2^16, Morlet, 128 scales, device resident, processed in
slabs of 128 signals (17 GB of W per slab, the slab buffer is re-used).  Prints throughput and checks
sampled rows against the oracle.  python tests/perf/config4_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pycwt_amd import _hip
from oracle import cwt_oracle as orc

nb_total, slab, N, rows = 1024, 128, 1 << 16, 128
m = orc.Mother(orc.MORLET, 6)
s0 = 2 / m.flambda(); dj = np.log2(N / s0) / (rows - 1)
sj = s0 * 2 ** (np.arange(rows) * dj)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
X = torch.randn(nb_total, N, dtype=torch.float64, device=dev, generator=g)
xh = torch.empty(slab, N, dtype=torch.complex128, device=dev)
W = torch.empty(slab, rows, N, dtype=torch.complex128, device=dev)
plan = _hip.Plan(N, 64, max_rows=slab * rows)
plan.set_stream(torch.cuda.current_stream().cuda_stream)

def run_spectra():                                    # rounds 1-2: spectra first, then the rows from the spectra alone
    for b0 in range(0, nb_total, slab):
        plan.fft_rows(X[b0:b0 + slab].data_ptr(), False, slab, N, N, xh.data_ptr())
        plan.transform_rows_batch(xh.data_ptr(), slab, N, 0, 6.0, 1.0, sj, W.data_ptr(), N, N)

def run():                                            # cwt_transform_batch: with the signals, time-compact rows go overlap-save
    for b0 in range(0, nb_total, slab):
        plan.transform_batch(X[b0:b0 + slab].data_ptr(), slab, N, N, 0, 6.0, 1.0, sj, xh.data_ptr(), W.data_ptr(), N, N)

for name, f in (("from the spectra (cwt_fft_rows + cwt_transform_rows_batch)", run_spectra), ("cwt_transform_batch", run)):
    f(); f(); torch.cuda.synchronize()                # second pass: the clock is up (tools/clock_ramp.py)
    t = time.perf_counter(); f(); torch.cuda.synchronize(); el = time.perf_counter() - t
    print(f"config 4 on 1 GPU, {name}: {nb_total} x 2^16 x {rows}: {el*1e3:.1f} ms -> {nb_total*N*rows/el/1e9:.1f} "
          f"GSamples*scales/s (split per slab {plan.last_split()})")
# parity: last slab holds signals 896..1023
sel = [0, 40, 90, 127]
for b in (0, 77, 127):
    ref = orc.cwt_rows(X[nb_total - slab + b].cpu().numpy(), 1.0, sj[sel], m)
    got = W[b, sel].cpu().numpy()
    err = (np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
    print(f"signal {nb_total - slab + b}: max row error {err:.2e}")
    assert err < 1e-8        # 1/100 of north_star's bar; the plan's default accuracy target is 1e-9
