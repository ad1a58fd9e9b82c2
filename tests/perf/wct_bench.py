"""BASELINE config 5 without the Monte-Carlo loop: pycwt_amd.xwt and pycwt_amd.wct of two N = 2^20 series (NumPy in, NumPy
out, PCIe included), and the reference's time for the same calls where it is mounted and small enough to wait for.
    python tests/perf/wct_bench.py [log2 N] [dj]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pycwt_amd

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dj = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
n = 1 << logn
rng = np.random.default_rng(55)
e = rng.standard_normal(n)
y1 = e + np.sin(2 * np.pi * np.arange(n) / 500.0)
y2 = 0.5 * np.roll(e, 3) + rng.standard_normal(n) + np.sin(2 * np.pi * np.arange(n) / 500.0 + 0.7)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t); del r
    return min(ts)


t_x = best(lambda: pycwt_amd.xwt(y1, y2, 1.0, dj))
t_w = best(lambda: pycwt_amd.wct(y1, y2, 1.0, dj, sig=False))
rows = pycwt_amd.xwt(y1, y2, 1.0, dj)[0].shape[0]
print(f"N = 2^{logn}, dj = {dj}: {rows} scales.  xwt {t_x * 1e3:.1f} ms ({rows * n * 16 / t_x / 1e9:.1f} GB/s of W12 to the host), "
      f"wct (sig=False) {t_w * 1e3:.1f} ms ({rows * n * 16 / t_w / 1e9:.1f} GB/s of WCT + angle to the host)")

def device_resident(f):
    r = f(); (r[0] if isinstance(r, tuple) else r).close()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t); (r[0] if isinstance(r, tuple) else r).close()
    return min(ts)


t_xd = device_resident(lambda: pycwt_amd.xwt_device(y1, y2, 1.0, dj))
t_wd = device_resident(lambda: pycwt_amd.wct_device(y1, y2, 1.0, dj))
print(f"device-resident results: xwt_device {t_xd * 1e3:.1f} ms, wct_device {t_wd * 1e3:.1f} ms (two uploads of {n * 8 / 1e6:.0f} MB included)")

if len(sys.argv) > 3:                                   # Monte-Carlo significance: draws per second for this scale grid
    mc = int(sys.argv[3])
    from pycwt_amd import wavelet as w
    m = pycwt_amd.Morlet(6)
    s0 = 2 * 1.0 / m.flambda()
    J = int(np.round(np.log2(n * 1.0 / s0) / dj))
    N, sj, *_ = w._mc_setup(m, 1.0, dj, s0, J)
    np.random.seed(3)
    pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=2, progress=False, cache=False)      # plans, tables

    def call(count, **kw):
        t = time.perf_counter()
        pycwt_amd.wct_significance(0.5, 0.4, 1.0, dj, s0, J, mc_count=count, progress=False, cache=False, **kw)
        return time.perf_counter() - t

    # A call = a fixed part (scratch of ~60 GB allocated at its first draw and freed at its end, the first draw's look at the
    # spectra, row tables of an accuracy target the plan has not seen) + draws.  Two calls of 2 and `mc` draws separate them.
    print(f"wct_significance for that grid: surrogates of {N} samples x {len(sj)} scales; per draw = (call of {mc} draws - call of 2) / {mc - 2}")
    for surr in ("reference", "ar1"):
        for rng_ in ("numpy", "device"):
            kw = dict(surrogates=surr, rng=rng_)
            call(2, **kw)
            t2, tm = call(2, **kw), call(mc, **kw)
            per = (tm - t2) / (mc - 2)
            print(f"   surrogates={surr!r:12s} rng={rng_!r:9s} {per * 1e3:7.1f} ms per draw, {max(t2 - 2 * per, 0.0):5.2f} s per call "
                  f"(the reference's default 300 draws: {(t2 - 2 * per + 300 * per):.0f} s on one GPU)")
