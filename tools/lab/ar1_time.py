import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pycwt_amd import _hip
import pycwt_amd
N = 6090214
plan = _hip.Plan(1 << 23, 64, max_rows=80)
e = _hip.DeviceBuffer((N + 8) * 8); x = _hip.DeviceBuffer(N * 8)
def t(f, reps=5):
    f(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3
print("random_normal %.3f ms" % t(lambda: plan.random_normal(1, 0, N + 3, 1.0, e.ptr)))
for g in (0.5, 0.9, 0.99):
    print("ar1_filter g=%.2f %.3f ms" % (g, t(lambda: plan.ar1_filter(e.ptr, 3, N, g, x.ptr))))
y = x.download(plan, (N,), np.float64)
print("finite", np.isfinite(y).all(), "lag1", np.corrcoef(y[:-1], y[1:])[0, 1])
# pieces of a draw
from pycwt_amd import wavelet as w
m = pycwt_amd.Morlet(6); dj = 0.25; s0 = 2 / m.flambda(); J = int(np.round(np.log2((1 << 20) / s0) / dj))
for surr in ("reference", "ar1"):
    for rng in ("numpy", "device"):
        kw = dict(progress=False, cache=False, surrogates=surr, rng=rng)
        pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=2, **kw)
        t0 = time.perf_counter(); pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=4, **kw); t4 = time.perf_counter() - t0
        t0 = time.perf_counter(); pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=12, **kw); t12 = time.perf_counter() - t0
        print(surr, rng, "fixed %.1f ms, per draw %.1f ms" % ((t4 - (t12 - t4) / 8 * 4) * 1e3, (t12 - t4) / 8 * 1e3))
