"""Where does the first draw of a Monte-Carlo call spend its time?  (wraps the shim's calls with timers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w, _hip

acc = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for nm in ("auto_tolerance", "set_tolerance", "transform", "forward_fft", "filter_rows", "fft_rows", "boxcar_scales", "coherence", "coherence_histogram", "sync", "transform_rows"):
    if hasattr(_hip.Plan, nm): timed(_hip.Plan, nm)
orig = w._coherence_on_device
def coh(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k)
    _hip.load()
    dt_ = time.perf_counter() - t
    print(f"      draw: {dt_ * 1e3:8.1f} ms   " + "  ".join(f"{k_} {v * 1e3:.1f}" for k_, v in acc.items() if v > 1e-3), flush=True); acc.clear()
    return r
w._coherence_on_device = coh
n = 1 << 20; dj = 0.25
m = pycwt_amd.Morlet(6)
s0 = 2 * 1.0 / m.flambda()
J = int(np.round(np.log2(n * 1.0 / s0) / dj))
np.random.seed(3)
for rep in range(2):
    for surr in ("reference", "ar1"):
        for rng_ in ("numpy", "device"):
            print(f"== rep {rep} {surr} {rng_}", flush=True)
            t = time.perf_counter()
            pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=4, progress=False, cache=False, surrogates=surr, rng=rng_)
            print(f"   call: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
