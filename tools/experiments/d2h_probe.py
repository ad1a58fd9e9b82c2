import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import pycwt_amd
from pycwt_amd import wavelet
x = np.random.default_rng(0).standard_normal(1 << 20)
f = lambda: pycwt_amd.cwt(x, 0.25, 0.25, wavelet="morlet")
def best(f, reps):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts), ts
print("discard result:", best(f, 5))
keep = []
def g():
    keep.append(f()[0])
    if len(keep) > 1: keep.pop(0)
print("keep previous result alive:", best(g, 5))
# pieces
plan = list(wavelet._plans.values())[0]
sj = f()[1]
t = time.perf_counter(); W = np.empty((sj.size, x.size), dtype=np.complex128); t1 = time.perf_counter() - t
t = time.perf_counter(); W[:] = 0; t2 = time.perf_counter() - t
print("np.empty %.2f ms, first touch by one thread %.1f ms" % (t1 * 1e3, t2 * 1e3))
del W
t = time.perf_counter(); out = plan.execute_host(x, 0, 6.0, 0.25, sj); t3 = time.perf_counter() - t
t = time.perf_counter(); del out; t4 = time.perf_counter() - t
print("execute_host %.1f ms, free of the result %.1f ms" % (t3 * 1e3, t4 * 1e3))
