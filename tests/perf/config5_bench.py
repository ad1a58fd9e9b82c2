"""BASELINE config 5 on ONE GPU: wavelet coherence of two N = 2^20 series (Morlet, dj = 1/12, default scale
grid: 229 scales), device resident; and the cost of ONE Monte-Carlo surrogate pair of its significance test
(series length 6 * s_max/dt -> transform length 2^23).  The 300 draws of the reference split over the ranks in
`pycwt_amd.parallel.wct_significance_sharded`.    python tests/perf/config5_bench.py [--mc]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd as pc
from pycwt_amd import wavelet as wv

N = 1 << 20
rng = np.random.default_rng(5)
y1 = pc.rednoise(N, 0.7, 1) if hasattr(pc, "rednoise") else rng.standard_normal(N)
y2 = 0.5 * y1 + rng.standard_normal(N)
def best_of(f, reps=4):
    out, times = None, []
    for _ in range(reps):
        t = time.perf_counter(); out = f(); times.append(time.perf_counter() - t)
    return out, min(times), max(times)

(WCT, aWCT, coi, freq, sig), lo, hi = best_of(lambda: pc.wct(y1, y2, 1.0, sig=False))
print(f"wct N=2^20: {WCT.shape[0]} scales, {lo*1e3:.0f}-{hi*1e3:.0f} ms end to end over 4 calls (host in, two "
      f"{WCT.nbytes/2**30:.1f} GiB result matrices out over PCIe into fresh NumPy arrays); coherence in "
      f"[{np.nanmin(WCT):.3f}, {np.nanmax(WCT):.3f}]")
(Wx, coi, freq, signif), lo, hi = best_of(lambda: pc.xwt(y1, y2, 1.0))
print(f"xwt N=2^20: {Wx.shape[0]} scales, {lo*1e3:.0f}-{hi*1e3:.0f} ms end to end over 4 calls "
      f"({Wx.nbytes/2**30:.1f} GiB complex result over PCIe)")
if "--mc" in sys.argv:
    m = pc.Morlet(6)
    s0 = 2 / m.flambda(); J = WCT.shape[0] - 1
    Nmc, sj, outside, rows_with_data, maxscale = wv._mc_setup(m, 1.0, 1 / 12, s0, J)
    t = time.perf_counter(); n1 = pc.rednoise(Nmc, 0.7, 1); n2 = pc.rednoise(Nmc, 0.6, 1)
    print(f"two surrogate series on the host: {time.perf_counter() - t:.2f} s")
    plan = wv._plan(wv._next_pow2(Nmc), 64, 0, sj.size)
    plan.set_option("profile", 1); plan.timings()
    wv._mc_histogram(1, 0.7, 0.6, 1.0, 1 / 12, sj, Nmc, outside, maxscale, m, 64, 0)
    tm = plan.timings(); plan.set_option("profile", 0)
    print("GPU time of one draw by kernel class (ms):", {k: round(v[0], 1) for k, v in tm.items()}, "sum",
          round(sum(v[0] for v in tm.values()), 1))
    times = []
    for draws in (2, 8):
        t = time.perf_counter()
        hist = wv._mc_histogram(draws, 0.7, 0.6, 1.0, 1 / 12, sj, Nmc, outside, maxscale, m, 64, 0)
        times.append(time.perf_counter() - t)
    per = (times[1] - times[0]) / 6
    print(f"Monte-Carlo significance: series length {Nmc} (transform length 2^{int(np.ceil(np.log2(Nmc)))}), "
          f"{hist.shape[0]} scales: 2 draws {times[0]:.2f} s, 8 draws {times[1]:.2f} s -> {per:.2f} s per "
          f"draw + {times[0] - 2 * per:.1f} s once (work matrices); 300 draws = {300 * per:.0f} s on one GPU, "
          f"{38 * per:.0f} s per rank on 8")
