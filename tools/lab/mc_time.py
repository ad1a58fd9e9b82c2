import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w, _hip
m = pycwt_amd.Morlet(6); dj = 0.25; dt = 1.0; s0 = 2 / m.flambda(); J = int(np.round(np.log2((1 << 20) / s0) / dj))
N, sj, outside, rows_with_data, maxscale = w._mc_setup(m, dt, dj, s0, J)
print("N", N, "rows", len(sj))
for surr in (False, True):
    orig = w._coherence_on_device
    times = []
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); times.append(time.perf_counter() - t0); return r
    w._coherence_on_device = timed
    t0 = time.perf_counter()
    w._mc_histogram(6, 0.5, 0.4, dt, dj, sj, N, outside, maxscale, m, 64, 0, ar1_surrogates=surr, rng="device", seed=5)
    tot = time.perf_counter() - t0
    w._coherence_on_device = orig
    plan = w._plan(1 << 23, 64, 0, len(sj))
    print("ar1" if surr else "white", "total %.1f ms; coherence calls:" % (tot * 1e3), ["%.1f" % (t * 1e3) for t in times], "tolerance", plan.tolerance(), plan.last_split())
