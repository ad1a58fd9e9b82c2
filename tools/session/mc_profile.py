import sys, os, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet as w
m = pycwt_amd.Morlet(6); dj = 0.25; n = 1 << 20
s0 = 2 / m.flambda(); J = int(np.round(np.log2(n / s0) / dj))
np.random.seed(3)
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=2, progress=False, cache=False)
pr = cProfile.Profile(); pr.enable()
pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=4, progress=False, cache=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
