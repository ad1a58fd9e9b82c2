"""pycwt_amd.cwt() at the reference's canonical size (504 samples, default grid): the whole call, the C call inside it,
and the Python around it (the C entry point replaced by a no-op).   python tests/perf/latency_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet


def per_call(f, n):
    f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e6


x = np.random.default_rng(0).standard_normal(504)
f = lambda: pycwt_amd.cwt(x, 0.25, 1 / 12, wavelet="morlet")
for _ in range(20):
    f()
whole = min(per_call(f, 500) for _ in range(5))
plan = next(iter(wavelet._plans.values()))
sj = f()[1]
inner = min(per_call(lambda: plan.execute_host(x, 0, 6.0, 0.25, sj), 500) for _ in range(5))
real = plan.lib.cwt_execute_host
plan.lib.cwt_execute_host = lambda *a: 0
stub = min(per_call(f, 2000) for _ in range(3))
plan.lib.cwt_execute_host = real
print(f"pycwt_amd.cwt 504 x {sj.size}: {whole:.1f} us per call; Plan.execute_host alone {inner:.1f} us; "
      f"Python with the C call stubbed {stub:.1f} us")

# the inverse of the same call (host W in, host vector out), next to the reference's NumPy expression (wavelet.py:169-170)
W, sj = pycwt_amd.cwt(x, 0.25, 1 / 12, wavelet="morlet")[:2]
g = lambda: pycwt_amd.icwt(W, sj, 0.25, 1 / 12, "morlet")
for _ in range(10):
    g()
inv = min(per_call(g, 300) for _ in range(3))
ref = lambda: (1 / 12 * np.sqrt(0.25) / (0.776 * np.pi ** -0.25)) * (np.real(W) / (np.ones([1, W.shape[1]]) * sj[:, None]) ** 0.5).sum(axis=0)
cpu = min(per_call(ref, 300) for _ in range(3))
print(f"pycwt_amd.icwt {W.shape[0]} x {W.shape[1]}: {inv:.1f} us per call; the reference's NumPy expression on this host: {cpu:.1f} us")
