import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, pycwt_amd
n = 1 << 20
rng = np.random.default_rng(55); e = rng.standard_normal(n)
y1 = e + np.sin(2 * np.pi * np.arange(n) / 500.0); y2 = 0.5 * np.roll(e, 3) + rng.standard_normal(n)
for _ in range(2):
    T, s = pycwt_amd.xwt_device(y1, y2, 1.0, 0.25); T.close()
t0 = time.perf_counter(); T, s = pycwt_amd.xwt_device(y1, y2, 1.0, 0.25); print("xwt_device %.1f ms" % ((time.perf_counter() - t0) * 1e3)); T.close()
pr = cProfile.Profile(); pr.enable(); T, s = pycwt_amd.xwt_device(y1, y2, 1.0, 0.25); pr.disable(); T.close()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
t0 = time.perf_counter(); D = pycwt_amd.wct_device(y1, y2, 1.0, 0.25); print("wct_device %.1f ms" % ((time.perf_counter() - t0) * 1e3)); D.close()
