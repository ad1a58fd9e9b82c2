import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w, _hip
n = 1 << 20
rng = np.random.default_rng(55)
e = rng.standard_normal(n)
y1 = e + np.sin(2 * np.pi * np.arange(n) / 500.0)
y2 = 0.5 * np.roll(e, 3) + rng.standard_normal(n)
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] = acc.get(label or name, 0) + (time.perf_counter() - t) * 1e3; return r
    setattr(obj, name, g)
wrap(w, "_normalised"); wrap(w, "_transform"); wrap(w, "_smooth_on_device"); wrap(w, "_coi")
wrap(_hip.DeviceBuffer, "__init__", "DeviceBuffer()"); wrap(_hip.DeviceBuffer, "download"); wrap(_hip.DeviceBuffer, "free"); wrap(_hip.DeviceBuffer, "upload")
wrap(_hip.Plan, "wct_products"); wrap(_hip.Plan, "wct_coherence")
for i in range(5):
    acc.clear()
    t = time.perf_counter(); r = pycwt_amd.wct(y1, y2, 1.0, 0.25, sig=False); tot = (time.perf_counter() - t) * 1e3
    del r
    print(f"wct {tot:.1f} ms:", {k: round(v, 1) for k, v in acc.items()})
