import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w
n = 1 << 20; dj = 0.25
m = pycwt_amd.Morlet(6)
s0 = 2 * 1.0 / m.flambda()
J = int(np.round(np.log2(n * 1.0 / s0) / dj))
np.random.seed(3)
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=2, progress=False, cache=False)
for rep in range(2):
    for surr in ("reference", "ar1"):
        for rng_ in ("numpy", "device"):
            kw = dict(mc_count=6, progress=False, cache=False, surrogates=surr, rng=rng_)
            t = time.perf_counter()
            pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, **kw)
            t = time.perf_counter() - t
            print(f"rep {rep} {surr:9s} {rng_:6s} {t / 6 * 1e3:7.1f} ms per draw", flush=True)
