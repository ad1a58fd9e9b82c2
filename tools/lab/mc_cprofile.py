"""cProfile of one Monte-Carlo call (2 draws) after a warm-up call: where the first draw's 1.5 s go."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd
n = 1 << 20; dj = 0.25
m = pycwt_amd.Morlet(6)
s0 = 2 * 1.0 / m.flambda()
J = int(np.round(np.log2(n * 1.0 / s0) / dj))
np.random.seed(3)
kw = dict(mc_count=2, progress=False, cache=False, surrogates="ar1", rng="device")
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, **kw)
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, **dict(kw, surrogates="reference"))
pr = cProfile.Profile(); pr.enable()
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
