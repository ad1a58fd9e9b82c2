import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w, helpers as h
n = 1 << 20
rng = np.random.default_rng(55)
e = rng.standard_normal(n)
y1 = e + np.sin(2 * np.pi * np.arange(n) / 500.0)
y2 = 0.5 * np.roll(e, 3) + rng.standard_normal(n)
acc = {}
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] = acc.get(label or name, 0) + (time.perf_counter() - t) * 1e3; return r
    setattr(mod, name, g)
wrap(w, "cwt_device"); wrap(w, "ar1"); wrap(w, "_normalised")
wrap(w.DeviceTransform, "W", "DeviceTransform.W")
from scipy.stats import chi2
for i in range(5):
    acc.clear()
    t = time.perf_counter(); r = pycwt_amd.xwt(y1, y2, 1.0, 0.25); tot = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); del r; fr = (time.perf_counter() - t) * 1e3
    print(f"xwt {tot:.1f} ms (freeing the result {fr:.1f}):", {k: round(v, 1) for k, v in acc.items()})
